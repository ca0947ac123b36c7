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
ap.add_argument("--valu-classes", action="store_true", help="split the vector instructions by issue cost class (tools/micro/valu_rates.hip)")
args = ap.parse_args()

# visits per wave trip (profiles/r05/census_hand_over_timers.txt, workload c2; the round-6 kernel's schedule is the same)
FREQ = {"prologue": 0.0, "trip": 1.0, "prio": 0.25, "service": 0.450, "publish": 0.30, "creation": 0.0584, "hand_out": 0.42, "walk": 0.9998, "crossing": 1.3083,
        "aim": 0.5998, "filter23": 0.1564, "search": 0.0192, "hit": 0.0023, "scatter": 0.9996, "liu": 0.9963, "hg": 0.9971, "rare": 0.0005}

# prop_device.hip.h: the region of a line is the region of the function it stands in (found by name, so that edits of the header do not move
# the map); inside propagate_through_layers the layer-crossing loop is a region of its own
FUNCTION_REGION = {
    "phase_ref_index": "creation", "group_velocity": "creation", "ice_factors": "creation", "generate_wavelength": "creation", "wavelength_bias": "creation",
    "step_direction": "creation", "work_direction": "creation", "photon_birth": "creation", "create_photon": "creation", "sph_dir_from_car": "creation",
    "table_bin_fraction": "creation", "table_value": "creation",
    "layer_lengths": None,                                    # walk or crossing -- by context
    "hg_cos": "hg", "liu_cos": "liu", "scatter_constants": "scatter", "scattering_cos": "scatter", "scatter_direction": "scatter",
    "abs_len_corr": "scatter", "apply_matrix": "scatter",
    "tilt_z_shift": "walk", "propagate_through_layers": "walk", "free_flight_bound": "walk", "free_flight_of": "walk",
    "dom_position": "search", "collide_with_string": "search", "find_collision_named": "search", "find_collision": "search",
    "save_hit_now": "search", "collide_with_string_keep": "search", "find_collisions_keep": "search",
    "segment_misses_string": "aim", "dom_search_needed": "filter23", "make_hit_record": "hit", "flush_hit_stubs": "hit"}
def device_regions():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "clsim_amd", "csrc", "prop_device.hip.h")
    out, current, in_walk, depth_at_loop = {}, "unmapped", False, None
    for n, text in enumerate(open(path).read().split("\n"), 1):
        m = re.match(r"^DM\s+[\w:<>\s\*&]+?\b(\w+)\(", text)
        if m and not text.startswith(" "):
            current = m.group(1)
        region = FUNCTION_REGION.get(current, "unmapped")
        if current == "propagate_through_layers":
            if "kCensusCrossing" in text:
                in_walk = True                                   # the loop's body: from its `while` line (one above) to its closing brace
                out[n - 1] = "crossing"
            if in_walk:
                region = "crossing"
                if text.strip() == "}":
                    in_walk = False
        if region != "unmapped" and region is not None:
            out[n] = region
    return out
DEVICE_REGIONS = device_regions()

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
        return DEVICE_REGIONS.get(line)
    return None

# vector instructions by issue cost (tools/micro/valu_rates.hip on this chip: cycles per wave64 instruction and SIMD in a stream of the same
# opcode; in the real kernel the differences are smaller -- profiles/r06/issue_cost_by_kind.txt): "fast" = the double-rate ones (2.2-2.6) with
# vector / literal / inline-constant sources, "sgpr" = the same with a scalar-register source (4.1), "slow" = everything else measured at 4.1-4.3
# (compares, conversions, min / max / med3, ldexp, shift-add, DPP, integer multiplies; unmeasured opcodes are counted here), "trans" = 8.2
FAST_OPS = ("v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mov_b32", "v_add_u32", "v_sub_u32",
            "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32")
def valu_class(op, text):
    base = re.sub(r"_e(32|64)$", "", op)
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", base): return "v_trans"
    if base in FAST_OPS and "dpp" not in text and "sdwa" not in text:
        operands = text.split(None, 1)[1] if " " in text else ""
        srcs = operands.split(",")[1:]
        return "v_sgpr" if any(re.match(r"\s*-?\|?(s\d+|s\[|vcc|exec|ttmp|m0)", o) for o in srcs) else "v_fast"
    return "v_slow"

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
    if k == "valu":
        cur["n"][valu_class(op, s.split(";")[0])] += 1
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
KINDS = ["valu", "salu", "exec", "branch", "smem", "wait", "nop", "lds", "vmem"] + (["v_fast", "v_sgpr", "v_slow", "v_trans"] if "--valu-classes" in sys.argv else [])
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
