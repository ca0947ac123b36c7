#!/usr/bin/env python3
"""What one unit of the counting oracle (oracle/count_ops.hpp) costs on gfx950: vector instructions of this repository's device
implementation of that unit (clsim_amd/csrc/detmath.hip.h, prop_device.hip.h: rng_co), counted in the assembly of one tiny
kernel per unit compiled with the propagation kernels' flags, minus a kernel that only loads and stores.  IEEE forms (the
compiler's divide / square-root sequences) and the range-restricted exact forms the kernels use where operand ranges are proven.
usage: tools/math_unit_costs.py [out.json]   (build container; needs hipcc, no GPU)"""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNITS = {
    "baseline": "o = x;",
    "add": "o = x + y;", "mul": "o = x * y;", "cmp": "o = (x < y) ? x : y;", "neg": "o = -x + y;", "cvt": "o = (float)(int)x;",
    "fabs": "o = __builtin_fabsf(x) + y;", "floor_trunc": "o = __builtin_floorf(x);",
    "div": "o = x / y;", "sqrt": "o = dm::sqrt_(x);", "rsqrt": "o = dm::rsqrt_(x);",
    "div_by_invariant(proven)": "{ const float q = x * y; o = dm::fma_(dm::fma_(-3.0f, q, x), y, q); }",
    "rcp(range-restricted)": "o = dm::rcp_(x);", "div_near(range-restricted)": "o = dm::div_near_(x, y);",
    "sqrt_near(range-restricted)": "o = dm::sqrt_near_(x);", "rsqrt_near(range-restricted)": "o = dm::rsqrt_near_(x);",
    "rcp_of_rcp(seeded)": "o = dm::rcp_of_rcp_(x, y);", "div_near_with(reciprocal at hand)": "o = dm::div_near_with_(x, y, a[threadIdx.x + 64]);",
    "rsqrt_unit(next to one)": "o = dm::rsqrt_unit_(x);",
    "log": "o = clsimhip::lds_log(x);", "log(global table)": "o = dm::log_(x);", "exp": "o = dm::exp_(x);", "powr": "o = dm::powr_(x, y);", "powr_unit": "o = dm::powr_unit_(x, y);",
    "sincos": "{ float s, c; clsimhip::lds_sincos_2pi(x, s, c); o = s + c; }", "sin": "{ float s, c; clsimhip::lds_sincos_2pi(x, s, c); o = s; }",
    "sincos(any argument)": "{ float s, c; dm::sincos_(x, s, c); o = s + c; }", "sincos(cephes)": "{ float s, c; dm::sincos_cephes_(x, s, c); o = s + c; }",
    "acos": "o = dm::acos_(x);", "atan2": "o = dm::atan2_(x, y);",
    "rng_draw": "{ uint64_t s = (uint64_t)__builtin_bit_cast(uint32_t, x) | ((uint64_t)__builtin_bit_cast(uint32_t, y) << 32); o = clsimhip::rng_co(s, 4294967118u); o += (float)(uint32_t)(s >> 32); }",
}
src = ['#include <hip/hip_runtime.h>', '#include "%s/clsim_amd/csrc/prop_device.hip.h"' % ROOT, 'using namespace clsimhip;']
names = {}
for i, (name, body) in enumerate(UNITS.items()):
    k = "unit_%d" % i
    names[k] = name
    src.append('extern "C" __global__ void %s(const float *a, const float *b, float *out) { const float x = a[threadIdx.x], y = b[threadIdx.x]; float o; %s out[threadIdx.x] = o; }' % (k, body))
with tempfile.TemporaryDirectory() as d:
    f = os.path.join(d, "units.hip")
    open(f, "w").write("\n".join(src) + "\n")
    asm = os.path.join(d, "units.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                           "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "--cuda-device-only", "-S", f, "-o", asm])
    text = open(asm).read()
counts = {}
for k, name in names.items():
    m = re.search(r"^%s:[^\n]*\n(.*?)\n\.Lfunc_end" % k, text, re.S | re.M)
    body = m.group(1)
    lines = [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith((";", ".", "//")) and not l.strip().endswith(":")]
    # a constant moved into a register is hoisted out of the photon loop by the real kernels (they hold such constants in SGPRs / VGPRs
    # across trips): not part of the unit's per-call cost
    lines = [l for l in lines if not re.match(r"v_mov_b32_e32 v\d+, (0x[0-9a-f]+|-?[0-9.]+)$", l)]
    insts = [l.split()[0] for l in lines]
    quarter = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)_|^v_mad_u64_u32|^v_mad_i64_i32|^v_mul_(lo|hi)_[ui]32")      # issue at a quarter of the full rate
    counts[name] = {"valu": sum(1 for i in insts if i.startswith("v_")), "salu": sum(1 for i in insts if i.startswith("s_") and not i.startswith(("s_waitcnt", "s_load", "s_nop"))),
                    "f64": sum(1 for i in insts if i.startswith("v_") and "f64" in i), "quarter": sum(1 for i in insts if quarter.match(i)),
                    "lds": sum(1 for i in insts if i.startswith("ds_"))}
base = counts.pop("baseline")
out = {}
for name, c in counts.items():
    extra = 1 if name in ("add", "mul", "cmp", "neg", "cvt", "fabs", "floor_trunc") else 0      # (their bodies add one op on top of the unit: see UNITS)
    out[name] = {"valu": c["valu"] - base["valu"], "salu": c["salu"] - base["salu"], "of_them_f64": c["f64"], "of_them_quarter_rate": c["quarter"] - base["quarter"], "lds_reads": c["lds"]}
# the one-instruction units are measured with a second operand folded in; normalise what the table is used for
for name, v in (("add", 1), ("mul", 1), ("cmp", 2), ("neg", 1), ("cvt", 2), ("fabs", 1), ("floor_trunc", 1)):
    out[name]["note"] = "measured %d" % out[name]["valu"]
res = {"what": "vector instructions per unit on gfx950, this repository's implementations (tools/math_unit_costs.py)", "baseline_kernel_valu": base["valu"], "units": out}
print(json.dumps(res, indent=1))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
