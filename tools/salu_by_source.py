#!/usr/bin/env python3
"""Scalar (and vector) instructions of the pooled propagation kernel PER WAVE TRIP, by origin.  BUILD CONTAINER TOOL.

Static part: the kernel's basic blocks from an annotated assembly listing (hipcc ... --cuda-device-only -gline-tables-only -S), each
instruction classified (vector ALU, scalar ALU, scalar branch, exec-mask bookkeeping, scalar memory load, s_waitcnt, s_nop, LDS, vector
memory).  Dynamic part: how often a block runs per wave trip, from the census build's region counts (tools/exp_pool_census.py:
visits per trip of the service block, creation, layer crossing, the filter levels, the searches, Liu / HG) -- a block is assigned
to a region by the source lines its instructions come from (the table REGIONS below; a block of nothing but math-library lines
inherits the region of the block before it).  The sum over blocks is checked against the PMC counters of the same kernel
(SQ_INSTS_SALU + SQ_INSTS_SMEM + SQ_INSTS_BRANCH, SQ_INSTS_VALU per wave trip): the model's total must come out near the measured one,
or the table is not to be trusted.

  tools/salu_by_source.py /tmp/pool_g.s [--symbol SYM] [--census workload=c2 key=value ...]"""
import argparse, collections, json, os, re, sys

ap = argparse.ArgumentParser()
ap.add_argument("asm")
ap.add_argument("--symbol", default="_ZN8clsimhip16prop_pool_kernelILi1ELb1ELb0ELb0ELb1ELb0EEEvNS_7KParamsE")
ap.add_argument("--measured", default=None, help="scalar_summary.json of tools/profile_scalar.sh (the check)")
ap.add_argument("--trips-per-wave", type=float, default=15814.0, help="census: wave trips per wave (C2: p50 15 814)")
ap.add_argument("--blocks", action="store_true", help="print every block with its region and weight")
args = ap.parse_args()

# visits per wave trip (profiles/r05/census_hand_over_timers.txt, workload c2; the round-6 kernel's schedule is the same)
FREQ = {"prologue": 0.0, "trip": 1.0, "prio": 0.25, "service": 0.450, "publish": 0.30, "creation": 0.0584, "hand_out": 0.42, "walk": 0.9998, "crossing": 1.3083,
        "aim": 0.5998, "filter23": 0.1564, "search": 0.0192, "hit": 0.0023, "scatter": 0.9996, "liu": 0.9963, "hg": 0.9971, "rare": 0.0005}

def region_of(file, line):
    if line == 0:
        return None                                          # (compiler-generated: no source line)
    if file.startswith("prop_pool_kernel"):
        if line < 161: return "prologue"
        if line <= 162: return "trip"
        if line <= 168: return "prio"                        # the priority switch: every 2^kPrioShift trips
        if line <= 175: return "trip"
        if line <= 224: return "publish" if 186 <= line <= 224 else "service"
        if line <= 350: return "creation"
        if line <= 388: return "hand_out"
        if line <= 434: return "trip"
        if line <= 476: return "search"
        if line <= 510: return "hit"
        if line <= 535: return "trip"
        return "prologue"
    if file.startswith("prop_device"):
        if 157 <= line <= 200: return "creation"            # group velocity, ice factors
        if 203 <= line <= 226: return None                   # layer_lengths: walk or crossing -- by context
        if 229 <= line <= 237: return "hg"
        if 239 <= line <= 246: return "liu"
        if 248 <= line <= 265: return "scatter"
        if 297 <= line <= 328: return "walk"                 # tilt
        if 331 <= line <= 375: return "creation"
        if 376 <= line <= 396: return "search"
        if 398 <= line <= 432: return "scatter"
        if 471 <= line <= 541: return "creation"
        if 596 <= line <= 604: return "crossing"
        if 544 <= line <= 625: return "walk"
        if 635 <= line <= 694: return "search"
        if 696 <= line <= 705: return "walk"
        if 717 <= line <= 746: return "aim"
        if 749 <= line <= 798: return "filter23"
        if 800 <= line <= 941: return "search"
        if line >= 1060: return "hit"
    return None

def kind(op, text):
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith(("s_load", "s_buffer_load")): return "smem"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc")): return "branch"
    if op.startswith("s_"):
        return "exec" if ("exec" in text or "saveexec" in op) else "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vmem"
    return "other"

asm = open(args.asm).read().split("\n")
start = [i for i, l in enumerate(asm) if l.startswith(args.symbol + ":")][0]
end = [i for i, l in enumerate(asm) if i > start and l.startswith(".Lfunc_end")][0]
files = {}
for l in asm:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
blocks, cur, loc = [], None, ("?", 0)
cur = {"label": "entry", "n": collections.Counter(), "votes": collections.Counter(), "unlikely": False}
blocks.append(cur)
for l in asm[start + 1:end]:
    s = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        loc = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = {"label": m.group(1), "n": collections.Counter(), "votes": collections.Counter(), "unlikely": False}
        blocks.append(cur)
        continue
    if not s or s.startswith((".", ";", "//")):
        continue
    op = s.split()[0]
    k = kind(op, s.split(";")[0])
    cur["n"][k] += 1
    r = region_of(*loc)
    if r:
        cur["votes"][r] += 1
# the loop's latch: the last block with a backward branch to the loop header; what the compiler placed behind it (blocks marked unlikely:
# IEEE fall-backs, the exits of rare paths) runs in about one trip in two thousand
labels, branches = [], []           # block labels in text order; (block index, target) of every branch
for l in asm[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels.append(m.group(1))
        continue
    m = re.search(r"\bs_c?branch\w*\s+(\.LBB\d+_\d+)\b", l)
    if m and labels:
        branches.append((len(labels) - 1, m.group(1)))
index = {lab: i for i, lab in enumerate(labels)}
# the main loop: the backward branch that spans most blocks
span, latch = 0, None
for at, target in branches:
    if target in index and index[target] <= at and at - index[target] > span:
        span, latch = at - index[target], labels[at]
after_latch = False
for b in blocks:
    b["cold"] = after_latch
    if b["label"] == latch:
        after_latch = True
# regions: majority of the classified lines; blocks without any inherit the previous block's
prev = "prologue"
for b in blocks:
    if b["votes"]:
        # the rarest region that holds at least a third of the classified instructions wins (a block that belongs to a rare region
        # also carries lines of its common surroundings, not the other way round)
        total = sum(b["votes"].values())
        cands = [r for r, v in b["votes"].items() if v * 3 >= total]
        b["region"] = min(cands, key=lambda r: FREQ[r])
    else:
        b["region"] = prev
    prev = b["region"]
    if b["cold"] and b["region"] not in ("hit", "prologue"):
        b["region"] = "rare"
KINDS = ["valu", "salu", "exec", "branch", "smem", "wait", "nop", "lds", "vmem"]
by_region = collections.defaultdict(lambda: collections.Counter())
for b in blocks:
    w = FREQ[b["region"]]
    for k in KINDS:
        by_region[b["region"]][k] += w * b["n"][k]
        by_region[b["region"]]["static_" + k] += b["n"][k]
if args.blocks:
    print("%-12s %-9s %6s | %s" % ("block", "region", "weight", " ".join("%5s" % k for k in KINDS)))
    for b in blocks:
        print("%-12s %-9s %6.3f | %s" % (b["label"], b["region"], FREQ[b["region"]], " ".join("%5d" % b["n"][k] for k in KINDS)))
    print()
print("per WAVE TRIP (static count x visits per trip); scalar = salu + exec-mask bookkeeping + branches + scalar loads")
print("%-10s %7s | %s | %7s" % ("region", "visits", " ".join("%7s" % k for k in KINDS), "scalar"))
tot = collections.Counter()
for r in sorted(by_region, key=lambda r: -sum(by_region[r][k] for k in ("salu", "exec", "branch", "smem"))):
    c = by_region[r]
    sc = c["salu"] + c["exec"] + c["branch"] + c["smem"]
    print("%-10s %7.4f | %s | %7.1f" % (r, FREQ[r], " ".join("%7.1f" % c[k] for k in KINDS), sc))
    for k in KINDS:
        tot[k] += c[k]
sc = tot["salu"] + tot["exec"] + tot["branch"] + tot["smem"]
print("%-10s %7s | %s | %7.1f" % ("total", "", " ".join("%7.1f" % tot[k] for k in KINDS), sc))
if args.measured and os.path.exists(args.measured):
    m = json.load(open(args.measured))["counters_per_launch"]
    trips = m["SQ_WAVES"] * args.trips_per_wave
    print("\nmeasured (rocprofv3 --pmc, %s): per wave trip (%.0f waves x %.0f trips): vector %.1f, scalar ALU %.1f, scalar loads %.1f, branches %.1f, LDS %.1f, vector memory %.1f"
          % (args.measured, m["SQ_WAVES"], args.trips_per_wave, m["SQ_INSTS_VALU"] / trips, m["SQ_INSTS_SALU"] / trips, m["SQ_INSTS_SMEM"] / trips,
             m["SQ_INSTS_BRANCH"] / trips, m["SQ_INSTS_LDS"] / trips, m["SQ_INSTS_VMEM"] / trips))
