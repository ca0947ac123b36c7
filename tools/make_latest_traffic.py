#!/usr/bin/env python3
"""profiles/latest_traffic.json from a tools/profile_round.sh summary (the file bench.py quotes `roofline.traffic` and
`roofline.valu` from, with the kernel name and the git revision the counters were taken at).
usage: make_latest_traffic.py gpurun_out/prof_<tag>/summary.json <git revision> [scalar-mix summary.json]
The in-tree library has to be the build that was profiled: its code hashes go into the file (tests/test_profile_freshness.py)."""
import json, os, subprocess, sys
d = json.load(open(sys.argv[1]))
rev = sys.argv[2]
kernel = [k for k in d["kernels"] if "prop_pool_kernel" in k or "prop_kernel<" in k][0]
c = d["prop_kernel_counters_per_launch"]
out = {
    "bytes_per_launch": int((c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024),
    "fetch_size_kb_per_launch": c["FETCH_SIZE"], "write_size_kb_per_launch": c["WRITE_SIZE"],
    "kernel": kernel.replace("void ", "").split("(")[0], "git_revision": rev, "kernel_ms": d["kernels"][kernel]["avg_ns"] / 1e6,
    "command": "rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path ; rocprofv3 --pmc WRITE_SIZE -- (same)   [tools/profile_round.sh, separate passes, no trace domains]",
    "source": "per-launch average over the launches rocprofv3 saw; counter rows of one dispatch summed over XCDs",
    "correction": "none applied: the guide's x2 FETCH_SIZE correction is calibrated for 16-B-per-lane coalesced streaming reads only; this kernel's accesses are scattered 4-64 B reads (work records, proximity map words, slice counters) and 8-B/64-B scattered writes, which the guide calls uncalibrated. The counters tally L2 fabric-side requests and include Infinity-Cache hits; the bunch's whole working set (48 MB steps + 64 MB work records + 61 MB proximity map + 15 MB hits) is Infinity-Cache resident, so real HBM traffic is far lower.",
    "interpretation": "about 170x the algorithmic 90 MB, and not on the critical path (0.26 TB/s of 8). Three sources: (1) every photon creation reads its step's 64-byte work record, and the records in flight per XCD (75 000 unit slots x 64 B = 4.8 MB) do not stay in the 4 MB L2: 2e8 photons x 64 B = 12.8 GB; (2) the second-level DOM proximity map (61 MB, one word per lookup, 1.3e8 lookups by the lanes that reach a string cylinder): 7.6 GB, measured by switching the level off (profiles/r02/v16_traffic_attribution_experiment.txt: 9.9 GB without it; non-temporal loads make it worse); (3) slice hand-offs through agent-scope (sc1) loads and write-through stores that bypass the XCD-local L2 by design (another XCD continues the stream): 16.8 M hand-offs x (2-3 line reads + 2 write-throughs) x 64 B = 3-4 GB + 1.2 GB of writes.",
    "sq_insts_valu_per_launch": c.get("SQ_INSTS_VALU"), "sq_insts_salu_per_launch": c.get("SQ_INSTS_SALU"),
    "valu_lane_utilisation": d.get("valu_lane_utilisation"),
    "valu_note": "separate --pmc pass of the same command (SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU)); a wave64 VALU operation occupies a SIMD for 2 cycles, so the issue-slot fraction is insts x 2 / (1024 SIMDs x 2.4e9 Hz x kernel time); scalar instructions compete for issue with the same weight (profiles/r02/v15_issue_cost_experiment.txt)",
}
# what was profiled, as machine code (tools/code_hash.py): every code object of the library, and the profiled kernel's own instructions
tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "code_hash.py")
out["code_hashes"] = [l.split()[0] for l in subprocess.run([sys.executable, tool], capture_output=True, text=True, check=True).stdout.splitlines() if l.strip()]
out["code_hashes_note"] = ("sha256[:16] of the .text of every gfx950 code object of clsim_amd/libclsimhip.so as profiled; tests/test_profile_freshness.py accepts "
                           "later edits of the kernel SOURCES while the library's hashes are these, or while the profiled kernel's own hash (kernel_hash) is")
rows = [l.split() for l in subprocess.run([sys.executable, tool, "--kernels"], capture_output=True, text=True, check=True).stdout.splitlines() if l.strip()]
names = subprocess.run(["c++filt"], input="\n".join(r[2] for r in rows), capture_output=True, text=True, check=True).stdout.splitlines()
for r, name in zip(rows, names):
    if name.replace("void ", "").split("(")[0] == out["kernel"]:
        out["kernel_hash"] = {"symbol": r[2], "sha16": r[0], "instructions": int(r[1]),
                              "note": "tools/code_hash.py --kernels: the profiled kernel's own instructions (addresses and branch targets taken out)"}
print(json.dumps(out, indent=1))
