#!/usr/bin/env python3
"""The "current numbers" table of DESIGN.md section 6, generated from the kept measurement files of the round (profiles/r06/final_*:
bench.py lines and rocprofv3 summaries taken at the shipped revision).  `--write` replaces the block between the numbers markers in
DESIGN.md; tests/test_design_numbers.py fails when DESIGN.md and the files disagree."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, "profiles", "r06")
ROWS = [("C2 = configs[1] (the headline; `python bench.py`)", "final_bench_default_line.json", "final_c2_pmc_summary.json"),
        ("C3 = configs[2], 10 M steps in two bunches (`--workload c3`)", "final_bench_c3.json", "final_c3_pmc_summary.json"),
        ("C5 = configs[4], flasher half (`--workload c5`)", "final_bench_c5.json", "final_c5_pmc_summary.json"),
        ("the reference's benchmark.py, steps born in HBM (`--workload benchmark`)", "final_bench_benchmark.json", None),
        ("the same through the reference's host API (`--workload benchmark-host`)", "final_bench_benchmark_host.json", None),
        ("C2 without `STOP_PHOTONS_ON_DETECTION` (`--keep-detected`)", "final_bench_c2_keep.json", None),
        ("C3 without `STOP_PHOTONS_ON_DETECTION`", "final_bench_c3_keep.json", None),
        ("C5 without `STOP_PHOTONS_ON_DETECTION`", "final_bench_c5_keep.json", None),
        ("table maker, 200×36×100×105 bins (`--workload tab`; tabulated photons/s)", "final_bench_tab.json", None),
        ("table maker with the impact-angle axis (`--workload tab5`)", "final_bench_tab5.json", None)]


def load(name):
    p = os.path.join(R, name)
    if not os.path.exists(p):
        return None
    text = open(p).read().strip()
    try:
        return json.loads(text)                     # a JSON document (rocprofv3 summaries, pretty-printed)
    except ValueError:
        return json.loads([l for l in text.splitlines() if l.startswith("{")][-1])       # a bench.py log: the line that is the result


def sci(v):
    m, e = ("%.2e" % v).split("e")
    return "%s·10%s" % (m, str(int(e)).translate(str.maketrans("0123456789-", "⁰¹²³⁴⁵⁶⁷⁸⁹⁻")))


def table():
    out = ["| workload | photons/s | kernel ms per launch | `useful_frac` of the vector peak | issue slots / lane use / `SQ_WAIT_ANY` share (rocprofv3 `--pmc`) |", "|---|---|---|---|---|"]
    for label, bench, pmc in ROWS:
        b = load(bench)
        if b is None:
            continue
        roof = b.get("roofline") or {}
        ms = roof.get("avg_kernel_ms") or b.get("kernel_ms_per_pass")
        valu = roof.get("valu") or {}
        useful = ("%.2f" % valu["useful_frac"]) if valu.get("useful_frac") else ("%.2f of the scattered-atomic rate" % roof["frac_of_scattered_rate"] if roof.get("frac_of_scattered_rate") else "")
        counters = ""
        p = load(pmc) if pmc else None
        if p:
            c = p["prop_kernel_counters_per_launch"]
            k = [x for x in p["kernels"] if "prop_pool_kernel" in x or "prop_kernel<" in x][0]
            t = p["kernels"][k]["avg_ns"] * 1e-9
            counters = "%.2f / %.2f / %.2f" % (c["SQ_INSTS_VALU"] * 2.0 / (1024 * 2.4e9 * t), p["valu_lane_utilisation"], c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"])
        extra = ""
        if b.get("host_path"):
            extra = "; through host buffers %s at %.0f %% device use" % (sci(b["host_path"]["value"]), 100 * b["host_path"]["device_utilization"])
        if b.get("host_path_copy"):
            extra += " (the caller's own copy of every result, the reference's `GetConversionResult()`: %s at %.0f %%)" % (
                sci(b["host_path_copy"]["value"]), 100 * b["host_path_copy"]["device_utilization"])
        if b.get("reference_figures"):
            extra = "; device utilisation %.2f" % b["reference_figures"]["DeviceUtilization"]
        out.append("| %s | **%s**%s | %s | %s | %s |" % (label, sci(b["value"]), extra, ("%.1f" % ms) if ms else "", useful, counters))
    d = load("final_bench_default_line.json")
    if d and d.get("cpu_baseline"):
        out.append("")
        out.append("CPU baseline of the same line: %s photons/s (%s, %d threads).  One-GPU rates of the N > 1 shards: §7." %
                   (sci(d["cpu_baseline"]["value"]), d["cpu_baseline"]["kind"], d["cpu_baseline"]["cores"]))
    return "\n".join(out)


BEGIN, END = "<!-- numbers:begin -->", "<!-- numbers:end -->"


def current_block():
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    return text[text.index(BEGIN) + len(BEGIN):text.index(END)].strip("\n")


if __name__ == "__main__":
    if "--write" in sys.argv:
        path = os.path.join(ROOT, "DESIGN.md")
        text = open(path).read()
        text = text[:text.index(BEGIN) + len(BEGIN)] + "\n" + table() + "\n" + text[text.index(END):]
        open(path, "w").write(text)
    print(table())
