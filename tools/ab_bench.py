#!/usr/bin/env python3
"""A/B of library builds on the GPU box: runs bench.py (kernel path only) once per library and workload, in the order
given, `--rounds` times over (interleaved, so that drift hits every build alike).

  tools/ab_bench.py --workloads c2,c3,c5 build_variants/base.so build_variants/new.so

Each library is a full libclsimhip.so (CLSIMHIP_LIB selects it); one JSON line per run goes to gpurun_out/ab_bench.jsonl and a
table of the best and median photons/s per build to stdout."""
import argparse
import json
import os
import statistics
import subprocess
import sys

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--workloads", default="c2")
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--bunch", type=int, default=0, help="steps per pass (default: the workload's own)")
ap.add_argument("--out", default="gpurun_out/ab_bench.jsonl")
args = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.dirname(os.path.join(root, args.out)), exist_ok=True)
results = {}
with open(os.path.join(root, args.out), "a") as log:
    for rnd in range(args.rounds):
        for wl in args.workloads.split(","):
            for lib in args.libs:
                env = dict(os.environ, CLSIMHIP_LIB=os.path.abspath(lib))
                cmd = [sys.executable, os.path.join(root, "bench.py"), "--workload", wl, "--steps", str(args.steps), "--warmup", "2",
                       "--no-cpu-baseline", "--no-host-path", "--no-table-maker"]
                if args.bunch:
                    cmd += ["--bunch", str(args.bunch)]
                p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
                line = [l for l in p.stdout.splitlines() if l.startswith("{")]
                if p.returncode != 0 or not line:
                    print("FAILED", lib, wl, p.stderr[-400:], flush=True)
                    continue
                r = json.loads(line[-1])
                rec = {"lib": os.path.basename(lib), "workload": wl, "round": rnd, "value": r["value"], "ms_per_step": r["ms_per_step"],
                       "kernel_ms": r.get("roofline", {}).get("avg_kernel_ms")}
                log.write(json.dumps(rec) + "\n")
                log.flush()
                results.setdefault((wl, os.path.basename(lib)), []).append(r["value"])
                print(rec, flush=True)
print("%-10s %-24s %12s %12s" % ("workload", "library", "best", "median"))
for (wl, lib), v in results.items():
    print("%-10s %-24s %12.4g %12.4g" % (wl, lib, max(v), statistics.median(v)))
