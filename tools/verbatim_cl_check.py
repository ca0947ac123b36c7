#!/usr/bin/env python3
"""Transcription check of the CPU oracle against the reference's OpenCL kernel text.  BUILD CONTAINER ONLY; TEST INFRASTRUCTURE.

oracle/clsim_oracle.c is a hand restatement of resources/kernels/{propagation_kernel,sparse_collision_kernel}.c.cl.  This
tool compiles those files VERBATIM, from where they lie under /root/reference, for x86-64 and runs them against the oracle:

  1. emit_program(): the run-time generated part of the OpenCL program (math preamble, mode #defines, wavelength generators,
     wavelength bias, medium functions, geometry constants) written from the oracle's table builders (oracle/builders.py) in
     the order the reference concatenates it (private/opencl/I3CLSimStepToPhotonConverterOpenCL.cxx:659-667).  The reference
     produces this text with C++ that needs IceTray + boost and cannot run here, so this part is this repository's
     restatement of those generators (each emitter cites the generator it follows); every float literal is written as a
     hexadecimal literal of the value the oracle's builder obtained from the reference's "%.10e" text round trip.
  2. the five kernel files are appended unchanged and compiled with ROCm's clang as OpenCL C 1.2 for x86_64
     (-O2 -ffp-contract=off -Dinline="static inline": C99 `inline` alone would leave saveHit & co. undefined).
  3. tools/cl_shim.cpp gives the ~40 OpenCL builtins the object leaves undefined: the math functions are oracle/oracle_math.h's
     (this repository's single-precision math definition -- the reference's would be its OpenCL runtime's, unpinned),
     conversions / clamp / mix / dot are written from the OpenCL 1.2 specification, get_global_id is the loop variable of a
     serial driver, async_work_group_copy is a memcpy, atom_inc an increment.
  4. the resulting propKernel runs over a step bunch, work item by work item, and its hit records and final RNG states are
     compared with oracle_propagate() on the same tables, steps and streams: bit for bit.

What this pins: the oracle's transcription of the two static kernel files (expression order, loop structure, branch
conditions, the layer walk, the collision search, saveHit).  What it cannot pin: the generated section and the builtins are
this repository's on both sides.  `--write-fixtures` stores the verbatim kernel's hit records under tests/golden/ (data:
inputs are regenerated from seeds, outputs are stored); tests/test_verbatim_cl.py compares oracle and HIP path with them.

usage: tools/verbatim_cl_check.py [--configs c1,mie,lea,flasher,photonics_mie,mie_history,mie_fixed_abs,lea_no_pancake] [--steps 4096] [--write-fixtures]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KERNELS = "/root/reference/resources/kernels"
CLANG = "/opt/rocm/lib/llvm/bin/clang"


def hexf(v):
    """C99 / OpenCL hexadecimal float literal of a binary32 value"""
    f = np.float32(v)
    assert np.isfinite(f), v
    return float(f).hex() + "f"


def fl(v):
    from oracle import builders as B
    return hexf(B.float_literal(v))


# ---------------------------------------------------------------------------------------------------------------------------
# the generated section
# ---------------------------------------------------------------------------------------------------------------------------
def emit_preamble(pancake, stop_detected=True, fixed_abs=None, history_n=0):
    """I3CLSimHelperMath.cxx:16-46 (single precision) + OpenCL.cxx:390-442 (mode #defines)"""
    s = ("#pragma OPENCL EXTENSION cl_khr_global_int32_base_atomics : enable\n"
         "#pragma OPENCL EXTENSION cl_khr_byte_addressable_store : enable\n"
         "typedef float floating_t;\ntypedef float2 floating2_t;\ntypedef float4 floating4_t;\n"
         "#define convert_floating_t convert_float\n#define ZERO 0.f\n#define ONE 1.f\n\n")
    if stop_detected:
        s += "#define STOP_PHOTONS_ON_DETECTION\n"
    if history_n:
        s += "#define SAVE_PHOTON_HISTORY\n#define NUM_PHOTONS_IN_HISTORY %d\n" % history_n
    if fixed_abs is not None:
        s += "#define PROPAGATE_FOR_FIXED_NUMBER_OF_ABSORPTION_LENGTHS %s\n" % fl(fixed_abs)
    if pancake != 1.0:
        s += "#define PANCAKE_FACTOR %s\n" % fl(pancake)
    return s


def emit_interpolated_distribution(name, gen):
    """random_value/I3CLSimRandomValueInterpolatedDistribution.cxx:177-336, constant spacing or (kind interp_x) with a table of
    x values (:203-211, :292-297)"""
    from oracle import builders as B
    yv, ycum = B.interp_dist_tables(gen)
    n = len(yv)
    p = "_" + name
    own_x = gen["kind"] == "interp_x"
    s = "#define %sNUM_DIST_ENTRIES %d\n" % (p, n)
    if own_x:
        s += "__constant float %sdistXValues[%sNUM_DIST_ENTRIES] = {%s};\n" % (p, p, ", ".join(hexf(v) for v in B.float_literals(gen["x"])))
    s += "__constant float %sdistYValues[%sNUM_DIST_ENTRIES] = {%s};\n" % (p, p, ", ".join(hexf(v) for v in yv))
    s += "__constant float %sdistYCumulativeValues[%sNUM_DIST_ENTRIES] = {%s};\n" % (p, p, ", ".join(hexf(v) for v in ycum))
    sp, first = (None, None) if own_x else (fl(gen["spacing"]), fl(gen["first"]))
    s += "inline float %s(RNG_ARGS);\ninline float %s(RNG_ARGS)\n{\n" % (name, name)
    s += "    const float randomNumber = RNG_CALL_UNIFORM_OC;\n"
    s += "    unsigned int k=0;\n    float this_acu = 0.f;\n"
    s += "    for (;;)\n    {\n        float next_acu = %sdistYCumulativeValues[k+1];\n" % p
    s += "        if (next_acu >= randomNumber) break;\n        this_acu = next_acu;\n        ++k;\n    }\n"
    s += "    const float b = %sdistYValues[k];\n" % p
    if own_x:
        s += "    const float x0 = %sdistXValues[k];\n" % p
        s += "    const float slope = (%sdistYValues[k+1]-b)/(%sdistXValues[k+1]-x0);\n" % (p, p)
    else:
        s += "    const float x0 = convert_float_rtz(k)*(%s) + (%s);\n" % (sp, first)
        s += "    const float slope = (%sdistYValues[k+1]-b)/(%s);\n" % (p, sp)
    s += "    const float dy = randomNumber-this_acu;\n"
    s += "    if ((b==0.f) && (slope==0.f))\n    {\n        return x0;\n    }\n"
    s += "    else if (b==0.f)\n    {\n        return x0 + sqrt(2.f*dy/slope);\n    }\n"
    s += "    else if (slope==0.f)\n    {\n        return x0 + dy/b;\n    }\n"
    s += "    else\n    {\n        return x0 + (sqrt(dy * (2.f*slope)/pown(b,2) + 1.f)-1.f)*b/slope;\n    }\n}\n"
    return s


def emit_wavelength_generators(generators):
    """I3CLSimHelperGenerateMediumPropertiesSource.cxx:392-461"""
    s = ""
    for i, g in enumerate(generators):
        name = "generateWavelength_%d" % i
        if g["kind"] in ("interp", "interp_x"):
            s += emit_interpolated_distribution(name, g) + "\n"
        elif g["kind"] == "const":                            # I3CLSimRandomValueConstant.cxx:75-103 (a fixed value)
            s += "inline float %s(RNG_ARGS);\ninline float %s(RNG_ARGS)\n{\n    return %s;\n}\n\n" % (name, name, fl(g["value"]))
        else:
            raise NotImplementedError(g["kind"])
    n = len(generators)
    s += "inline float generateWavelength(uint number, RNG_ARGS);\ninline float generateWavelength(uint number, RNG_ARGS)\n{\n"
    if n == 0:
        s += "    return 0.f;\n}\n"
    elif n == 1:
        s += "    return generateWavelength_0(RNG_ARGS_TO_CALL);\n}\n"
    else:
        s += "    if (number==0) {\n        return generateWavelength_0(RNG_ARGS_TO_CALL);\n"
        for i in range(1, n):
            s += "    } else if (number==%d) {\n        return generateWavelength_%d(RNG_ARGS_TO_CALL);\n" % (i, i)
        s += "    } else {\n        return 0.f;\n    }\n}\n"
    return s + "\n"


def emit_function_from_table(name, tab):
    """function/I3CLSimFunctionFromTable.cxx:167-300, float data"""
    n = len(tab["values"])
    s = "__constant float %s_data[%d] = {%s};\n" % (name, n, ", ".join(fl(v) for v in tab["values"]))
    h = name + "_getInterpolationBinAndFraction"
    s += "inline void %s(float wavelength, int *bin, float *fraction);\n" % h
    s += "inline void %s(float wavelength, int *bin, float *fraction)\n{\n    float fbin;\n" % h
    s += "    *fraction = modf((wavelength - %s)/%s, &fbin);\n    int ibin=(int)fbin;\n" % (fl(tab["start"]), fl(tab["step"]))
    s += "    if ((ibin<0) || ((ibin==0) && (*fraction<0))) {\n        ibin=0;\n        *fraction=0.f;\n"
    s += "    } else if (ibin>=%d-1) {\n        ibin=%d-2;\n        *fraction=1.f;\n    }\n    *bin = ibin;\n}\n" % (n, n)
    s += "inline float %s(float wavelength);\ninline float %s(float wavelength)\n{\n    int bin; float fraction;\n" % (name, name)
    s += "    %s(wavelength, &bin, &fraction);\n    return mix(%s_data[bin], %s_data[bin+1], fraction);\n}\n" % (h, name, name)
    return s


def emit_function_from_table_16bit(name, tab_start, tab_step, values):
    """function/I3CLSimFunctionFromTable.cxx:183-207, 262-277: values stored as 16-bit fractions of their range"""
    from oracle import builders as B
    lo, hi, q = B.quantize_table(values)
    n = len(q)
    d = name + "_data"
    s = "#define %s_SMALLEST_ENTRY %s \n#define %s_LARGEST_ENTRY %s \n" % (d, hexf(lo), d, hexf(hi))
    s += "__constant unsigned short %s[%d] = {%s};\n" % (d, n, ", ".join("%d" % int(v) for v in q))
    h = name + "_getInterpolationBinAndFraction"
    s += "inline void %s(float wavelength, int *bin, float *fraction);\n" % h
    s += "inline void %s(float wavelength, int *bin, float *fraction)\n{\n    float fbin;\n" % h
    s += "    *fraction = modf((wavelength - %s)/%s, &fbin);\n    int ibin=(int)fbin;\n" % (fl(tab_start), fl(tab_step))
    s += "    if ((ibin<0) || ((ibin==0) && (*fraction<0))) {\n        ibin=0;\n        *fraction=0.f;\n"
    s += "    } else if (ibin>=%d-1) {\n        ibin=%d-2;\n        *fraction=1.f;\n    }\n    *bin = ibin;\n}\n" % (n, n)
    s += "inline float %s(float wavelength);\ninline float %s(float wavelength)\n{\n    int bin; float fraction;\n" % (name, name)
    s += "    %s(wavelength, &bin, &fraction);\n" % h
    s += ("    return mix(convert_float(%s[bin])  *((%s_LARGEST_ENTRY-%s_SMALLEST_ENTRY)/65535.f) + %s_SMALLEST_ENTRY,\n"
          "               convert_float(%s[bin+1])*((%s_LARGEST_ENTRY-%s_SMALLEST_ENTRY)/65535.f) + %s_SMALLEST_ENTRY,\n"
          "               fraction);\n}\n" % (d, d, d, d, d, d, d, d))
    return s


def emit_layer_switch(name, n_layers):
    """MediumPropertiesSource.cxx:91-125: one function per layer behind a switch (identical function objects would share
    one; the tables of different layers differ)"""
    s = "inline float %s(unsigned int layer, float wavelength);\ninline float %s(unsigned int layer, float wavelength)\n{\n    switch(layer)\n    {\n" % (name, name)
    for i in range(n_layers):
        s += "        case %d: return %s_func%d(wavelength);\n" % (i, name, i)
    s += "        default: return 0.;\n    }\n}\n\n"
    return s


def emit_refindex(name, n, g, mode):
    """function/I3CLSimFunctionRefIndexIceCube.cxx:128-180"""
    from oracle import builders as B
    s = "inline float %s(float wlen);\ninline float %s(float wlen)\n{\n" % (name, name)
    for i in range(5):
        s += "    const float n%d = %s;\n" % (i, fl(n[i]))
    if mode == "group":
        for i in range(5):
            s += "    const float g%d = %s;\n" % (i, fl(g[i]))
    s += "    const float x = wlen/%s;\n    const float np = n0 + x*(n1 + x*(n2 + x*(n3 + x*n4)));\n" % fl(B.MICROMETER)
    if mode == "phase":
        s += "    return np;\n}\n"
    else:
        s += "    const float np_corr = g0 + x*(g1 + x*(g2 + x*(g3 + x*g4)));\n    return np*np_corr;\n}\n"
    return s


def emit_refindex_derivative(name, n):
    """function/I3CLSimFunctionRefIndexIceCube.cxx:182-231, mode "phase": the text GetOpenCLFunctionDerivative prints"""
    from oracle import builders as B
    s = "inline float %s(float wlen);\ninline float %s(float wlen)\n{\n" % (name, name)
    for i in range(1, 5):
        s += "    const float n%d = %s;\n" % (i, fl(n[i]))
    s += "    const float x = wlen/%s;\n" % fl(B.MICROMETER)
    s += "    const float dnp = (n1 + x*(2.f*n2 + x*(3.f*n3 + x*4.f*n4)))/%s;\n    return dnp;\n}\n" % fl(B.MICROMETER)
    return s


def emit_single_layer_wrapper(name):
    """MediumPropertiesSource.cxx:91-125 when every layer has the same function object"""
    return ("#define FUNCTION_%s_DOES_NOT_DEPEND_ON_LAYER\n"
            "inline float %s(unsigned int layer, float wavelength);\n"
            "inline float %s(unsigned int layer, float wavelength)\n{\n    return %s_func0(wavelength);\n}\n\n" % (name, name, name, name))


def emit_medium(m):
    """I3CLSimHelperGenerateMediumPropertiesSource.cxx:207-389 (+ _Optimizers.cxx:123-250)"""
    from oracle import builders as B
    s = "#define MEDIUM_LAYERS %d\n" % m["num_layers"]
    s += "#define MEDIUM_MIN_WLEN %s\n#define MEDIUM_MAX_WLEN %s\n" % (fl(m["min_wlen"]), fl(m["max_wlen"]))
    s += "#define MEDIUM_MIN_RECIP_WLEN %s\n#define MEDIUM_MAX_RECIP_WLEN %s\n" % (fl(1. / m["max_wlen"]), fl(1. / m["min_wlen"]))
    s += "#define MEDIUM_LAYER_BOTTOM_POS %s\n#define MEDIUM_LAYER_THICKNESS  %s\n\n" % (fl(m["layers_z_start"]), fl(m["layers_height"]))
    # phase refractive index: one RefIndexIceCube object for all layers (the dispersion function the reference also emits
    # is not called by the kernel when a group-index override exists: left out)
    if "phase_table" in m:
        s += emit_function_from_table("getPhaseRefIndex_func0", m["phase_table"]) + emit_single_layer_wrapper("getPhaseRefIndex")
    else:
        s += emit_refindex("getPhaseRefIndex_func0", m["n"], m["g"], "phase") + emit_single_layer_wrapper("getPhaseRefIndex")
    if m.get("group_from_dispersion"):
        # no group refractive index override: MediumPropertiesSource.cxx:274-308, with the derivative of the phase index
        # (GenerateLayeredWlenDependentFunctions(..., "getPhaseRefIndex", "getDispersion"), :230-235)
        s += emit_refindex_derivative("getDispersion_func0", m["n"]) + emit_single_layer_wrapper("getDispersion")
        s += "#ifdef FUNCTION_getPhaseRefIndex_DOES_NOT_DEPEND_ON_LAYER\n#define FUNCTION_getGroupVelocity_DOES_NOT_DEPEND_ON_LAYER\n"
        s += "#define FUNCTION_getGroupRefIndex_DOES_NOT_DEPEND_ON_LAYER\n#endif\n"
        s += "inline float getGroupVelocity(unsigned int layer, float wavelength);\n"
        s += "inline float getGroupVelocity(unsigned int layer, float wavelength)\n{\n    const float c_light = %s;\n" % fl(B.C_LIGHT)
        s += "#ifdef USE_NATIVE_MATH\n    const float n_inv = native_recip(getPhaseRefIndex(layer, wavelength));\n#else\n"
        s += "    const float n_inv = 1.f/getPhaseRefIndex(layer, wavelength);\n#endif\n"
        s += "    const float y = getDispersion(layer, wavelength);\n    return c_light * (1.0f + y*wavelength*n_inv) * n_inv;\n}\n\n"
        s += "inline float getGroupRefIndex(unsigned int layer, float wavelength);\n"
        s += "inline float getGroupRefIndex(unsigned int layer, float wavelength)\n{\n    const float c_light = %s;\n" % fl(B.C_LIGHT)
        s += "    const float groupvel = getGroupVelocity(layer, wavelength);\n    return c_light / groupvel;\n}\n\n"
    elif "group_table" in m:
        s += emit_function_from_table("getGroupRefIndex_func0", m["group_table"]) + emit_single_layer_wrapper("getGroupRefIndex")
    else:
        s += emit_refindex("getGroupRefIndex_func0", m["n"], m["g"], "group") + emit_single_layer_wrapper("getGroupRefIndex")
    if not m.get("group_from_dispersion"):
        s += "#ifdef FUNCTION_getGroupRefIndex_DOES_NOT_DEPEND_ON_LAYER\n#define FUNCTION_getGroupVelocity_DOES_NOT_DEPEND_ON_LAYER\n#endif\n"
        s += "inline float getGroupVelocity(unsigned int layer, float wavelength);\n"
        s += "inline float getGroupVelocity(unsigned int layer, float wavelength)\n{\n    const float c_light = %s;\n" % fl(B.C_LIGHT)
        s += "    const float n_group = getGroupRefIndex(layer, wavelength);\n    return c_light / n_group;\n}\n\n"
    if m["len_mode"] == "icecube":
        assert m["num_layers"] > 1
        nl = m["num_layers"]
        # _Optimizers.cxx:195-250
        s += "__constant float getScatteringLength_b400[%d] = {%s};\n" % (nl, ", ".join(fl(v) for v in m["b400"]))
        s += "inline float getScatteringLength(unsigned int layer, float wlen);\n"
        s += "inline float getScatteringLength(unsigned int layer, float wlen)\n{\n    const float alpha = %s;\n" % fl(m["alpha"])
        s += "    return %s/( getScatteringLength_b400[layer] * powr(wlen*%s, -alpha) );\n}\n\n" % (fl(1.0), fl(1. / (400. * B.NANOMETER)))
        # _Optimizers.cxx:123-190
        s += "__constant float getAbsorptionLength_aDust400[%d] = {%s};\n" % (nl, ", ".join(fl(v) for v in m["aDust400"]))
        s += "__constant float getAbsorptionLength_deltaTau[%d] = {%s};\n" % (nl, ", ".join(fl(v) for v in m["deltaTau"]))
        s += "inline float getAbsorptionLength(unsigned int layer, float wlen);\n"
        s += "inline float getAbsorptionLength(unsigned int layer, float wlen)\n{\n"
        for k in ("kappa", "A", "B", "D", "E"):
            s += "    const float %s = %s;\n" % (k, fl(m[k]))
        s += "    const float x = wlen/%s;\n" % fl(B.NANOMETER)
        s += ("    return %s/( (D*getAbsorptionLength_aDust400[layer]+E) * powr(x, -kappa)  +  A*exp(-B/x) * "
              "(1.f + 0.01f*getAbsorptionLength_deltaTau[layer]) );\n}\n\n" % fl(1.0))
    elif m["len_mode"] == "constant":
        # one FunctionConstant per slot (FunctionConstant.cxx:81-100) behind the generic layer switch
        # (MediumPropertiesSource.cxx:91-125); the optimisers decline for a single layer
        for name, key in (("getScatteringLength", "sca_const"), ("getAbsorptionLength", "abs_const")):
            vals = list(m[key])
            distinct = []
            which = []
            for v in vals:
                if v not in distinct:
                    distinct.append(v)
                which.append(distinct.index(v))
            for i, v in enumerate(distinct):
                s += "inline float %s_func%d(float wavelength);\ninline float %s_func%d(float wavelength)\n{\n    return %s;\n}\n" % (name, i, name, i, fl(v))
            if len(distinct) == 1:
                s += emit_single_layer_wrapper(name)
            else:
                s += "inline float %s(unsigned int layer, float wavelength);\ninline float %s(unsigned int layer, float wavelength)\n{\n    switch(layer)\n    {\n" % (name, name)
                for i, w in enumerate(which):
                    s += "        case %d: return %s_func%d(wavelength);\n" % (i, name, w)
                s += "        default: return 0.;\n    }\n}\n\n"
    elif m["len_mode"] == "table":
        tb = m["table"]
        assert tb["store16"]
        for name, key in (("getScatteringLength", "sca"), ("getAbsorptionLength", "abs")):
            for i in range(m["num_layers"]):
                s += emit_function_from_table_16bit("%s_func%d" % (name, i), tb["start"], tb["step"], tb[key][i])
            s += emit_layer_switch(name, m["num_layers"])
    else:
        raise NotImplementedError(m["len_mode"])
    # scattering angle (random_value/I3CLSimRandomValueMixed.cxx:115-157, SimplifiedLiu.cxx:64-88, HenyeyGreenstein.cxx:69-92)
    sc = m["scat"]
    g = sc["mean_cos"]

    def liu(name, args, u):
        return ("inline float %s(%s);\ninline float %s(%s)\n{\n    const float beta = %s;\n"
                "    return clamp(2.f * powr((%s), beta) - 1.f, -1.f, 1.f);\n}\n" % (name, args, name, args, fl((1. - g) / (1. + g)), u))

    def hg(name, args, u):
        return ("inline float %s(%s);\ninline float %s(%s)\n{\n    const float g = %s;\n    const float g2 = %s;\n"
                "    const float s = 2.f*(%s)-1.f;\n    const float ii = ((1.f - g2)/(1.f + g*s));\n"
                "    return clamp((1.f + g2 - ii*ii) / (2.f*g), -1.f, 1.f);\n}\n" % (name, args, name, args, fl(g), fl(g * g), u))
    if sc["kind"] == "mixed":
        f = sc["fraction"]
        s += liu("makeScatteringCosAngle_mix1", "float rrrr__", "rrrr__") + hg("makeScatteringCosAngle_mix2", "float rrrr__", "rrrr__")
        s += "inline float makeScatteringCosAngle(RNG_ARGS);\ninline float makeScatteringCosAngle(RNG_ARGS)\n{\n"
        s += "    const float rr = RNG_CALL_UNIFORM_CO;\n    if (rr < %s)\n    {\n        return makeScatteringCosAngle_mix1(rr/%s);\n    }\n" % (fl(f), fl(f))
        s += "    else\n    {\n        return makeScatteringCosAngle_mix2((1.f-rr)/%s);\n    }\n}\n\n" % fl(1. - f)
    elif sc["kind"] == "hg":
        s += hg("makeScatteringCosAngle", "RNG_ARGS", "RNG_CALL_UNIFORM_CO") + "\n"
    else:
        s += liu("makeScatteringCosAngle", "RNG_ARGS", "RNG_CALL_UNIFORM_CO") + "\n"
    # directional absorption length correction (ScalarFieldAnisotropyAbsLenScaling.cxx:92-140 / ScalarFieldConstant.cxx:61-80)
    if "aniso" in m:
        c = B.aniso_constants(m["aniso"])
        s += "inline float getDirectionalAbsLenCorrFactor(float4 vec);\ninline float getDirectionalAbsLenCorrFactor(float4 vec)\n{\n"
        s += "    const float4 l  = (float4)(%s, %s, %s, 0.f);\n" % tuple(fl(v) for v in c["l"])
        s += "    const float4 rl = (float4)(%s, %s, %s, 0.f);\n" % tuple(fl(v) for v in c["rl"])
        s += "    const float4 n = (float4)\n        (\n         (%s*vec.x)+(%s*vec.y),\n         (%s*vec.x)+(%s*vec.y),\n         vec.z,\n         0.f\n        );\n" % (
            fl(c["azx"]), fl(c["azy"]), fl(-c["azy"]), fl(c["azx"]))
        s += "    const float4 s=n*n;\n    const float nB = dot(s,rl);\n    const float An = dot(s,l);\n    return 2.f/((%s-nB)*An);\n}\n\n" % fl(c["B2"])
    else:
        s += emit_scalar_field_constant("getDirectionalAbsLenCorrFactor", 1.0)
    # direction transforms (VectorTransformMatrix.cxx:101-135 / VectorTransformConstant.cxx:58-74)
    for key, name in (("pre", "transformDirectionPreScatter"), ("post", "transformDirectionPostScatter")):
        s += "inline void %s(float4 *vec);\ninline void %s(float4 *vec)\n{\n" % (name, name)
        if key in m:
            mat = np.asarray(m[key]["matrix"], dtype=np.float64)
            s += "    *vec = (float4)\n    (\n"
            for i in range(3):
                s += "        (%s*(*vec).x)+(%s*(*vec).y)+(%s*(*vec).z),\n" % (fl(mat[i, 0]), fl(mat[i, 1]), fl(mat[i, 2]))
            s += "        (*vec).w\n    );\n"
            if m[key]["renormalize"]:
                s += "    const float norm = rsqrt((*vec).x*(*vec).x + (*vec).y*(*vec).y + (*vec).z*(*vec).z);\n    (*vec).xyz = (*vec).xyz*norm;\n"
            s += "}\n\n"
        else:
            s += "    return;\n}\n\n"
    # ice tilt (ScalarFieldIceTiltZShift.cxx:145-213 / ScalarFieldConstant)
    if "tilt" in m:
        tl = m["tilt"]
        first_z, dz = B.tilt_spacing(tl["zcoords"])
        d = "getTiltZShift_data"
        nd, nz = len(tl["distances"]), len(tl["zcoords"])
        s += "#define %s_numDistances  %d\n#define %s_numZCoords    %d\n" % (d, nd, d, nz)
        s += "#define %s_firstZCoord   %s\n#define %s_zCoordSpacing %s\n" % (d, fl(first_z), d, fl(dz))
        s += "__constant float %s_distancesFromOriginAlongTilt[%s_numDistances] = {%s};\n" % (d, d, ", ".join(fl(v) for v in tl["distances"]))
        s += "__constant float %s_zCorrections[%s_numDistances*%s_numZCoords] = {%s};\n" % (
            d, d, d, ", ".join(fl(v) for v in np.asarray(tl["zcorr"], dtype=np.float64).ravel()))
        s += "inline float getTiltZShift(float4 vec);\ninline float getTiltZShift(float4 vec)\n{\n"
        s += "    const float z_rescaled = (vec.z-%s_firstZCoord)/%s_zCoordSpacing;\n" % (d, d)
        s += "    const int k = min(max(convert_int_rtn(z_rescaled), 0), %s_numZCoords-2);\n" % d
        s += "    const float fraction_z_above = z_rescaled-convert_float(k);\n    const float fraction_z_below = 1.-fraction_z_above;\n"
        s += "    const float nr = %s*vec.x + %s*vec.y;\n" % (fl(np.cos(tl["azimuth"])), fl(np.sin(tl["azimuth"])))
        s += "    for(int j=1; j<%s_numDistances; j++)\n    {\n        const float thisDist = %s_distancesFromOriginAlongTilt[j];\n" % (d, d)
        s += "        if((nr<thisDist) || (j==%s_numDistances-1))\n        {\n" % d
        s += "            const float previousDist = %s_distancesFromOriginAlongTilt[j-1];\n" % d
        s += "            const float thisDistanceBinWidth = thisDist - previousDist;\n"
        s += "            const float frac_at_lower = (thisDist - nr    )/thisDistanceBinWidth;\n            const float frac_at_upper = 1.-frac_at_lower;\n"
        s += ("            const float val_at_lower = (%s_zCorrections[(j-1)*%s_numZCoords + k+1]*fraction_z_above + "
              "%s_zCorrections[(j-1)*%s_numZCoords + k]*fraction_z_below);\n" % (d, d, d, d))
        s += ("            const float val_at_upper = (%s_zCorrections[j    *%s_numZCoords + k+1]*fraction_z_above + "
              "%s_zCorrections[j    *%s_numZCoords + k]*fraction_z_below);\n" % (d, d, d, d))
        s += "            return (val_at_upper * frac_at_upper + val_at_lower * frac_at_lower);\n        }\n    }\n}\n\n"
    else:
        s += emit_scalar_field_constant("getTiltZShift", 0.0)
    return s


def emit_scalar_field_constant(name, value):
    """function/I3CLSimScalarFieldConstant.cxx:61-80"""
    return ("inline float %s(float4 vec);\n\n#define %s_IS_CONSTANT %s\ninline float %s(float4 vec)\n{\n    return %s;\n}\n\n"
            % (name, name, fl(value), name, fl(value)))


def emit_geometry(geo):
    """I3CLSimHelperGenerateGeometrySource.cxx:619-700, 1137-1272: the values are those of oracle/builders.py: build_geometry,
    which already went through the literal round trip"""
    def arr(ctype, name, size, vals, fmt):
        return "__constant %s %s[%s] = {%s};\n" % (ctype, name, size, ", ".join(fmt(v) for v in vals))
    ints = lambda v: "%d" % int(v)
    u16 = lambda v: ("0xFFFF" if int(v) == 0xFFFF else "%d" % int(v))
    s = "#define GEO_MAX_DOM_INDEX %d\n" % geo["max_dom_index"]
    s += "#define GEO_DOM_POS_MAX_ABS_X_MULTIPLIER_IN_TEMPLATE %s\n#define GEO_DOM_POS_MAX_ABS_Y_MULTIPLIER_IN_TEMPLATE %s\n" % (
        hexf(geo["dom_mul_x"]), hexf(geo["dom_mul_y"]))
    s += "#define GEO_DOM_POS_NUM_FLAT_LIST_ENTRIES %d\n" % len(geo["dom_tx"])
    s += arr("short", "geoDomPosTemplatePositionsX_flat", "GEO_DOM_POS_NUM_FLAT_LIST_ENTRIES", geo["dom_tx"], ints)
    s += arr("short", "geoDomPosTemplatePositionsY_flat", "GEO_DOM_POS_NUM_FLAT_LIST_ENTRIES", geo["dom_ty"], ints)
    s += arr("float", "geoDomPosTemplatePositionsZ_flat", "GEO_DOM_POS_NUM_FLAT_LIST_ENTRIES", geo["dom_tz"], hexf)
    s += "#define GEO_DOM_POS_NUM_STRINGS %d\n" % geo["num_strings"]
    s += arr("unsigned int", "geoDomPosStringStartIndexInTemplateDomList", "GEO_DOM_POS_NUM_STRINGS", geo["dom_start"], ints)
    s += arr("float", "geoDomPosStringMeanPosX", "GEO_DOM_POS_NUM_STRINGS", geo["dom_meanx"], hexf)
    s += arr("float", "geoDomPosStringMeanPosY", "GEO_DOM_POS_NUM_STRINGS", geo["dom_meany"], hexf)
    sig = "inline void geometryGetDomPosition(unsigned short stringNum, unsigned short domNum, floating_t *domPosX, floating_t *domPosY, floating_t *domPosZ)"
    s += sig + ";\n" + sig + "\n{\n"
    s += "    const unsigned int index = geoDomPosStringStartIndexInTemplateDomList[stringNum]+convert_uint(domNum);\n"
    s += "    *domPosX = convert_floating_t(geoDomPosTemplatePositionsX_flat[index])*GEO_DOM_POS_MAX_ABS_X_MULTIPLIER_IN_TEMPLATE + geoDomPosStringMeanPosX[stringNum];\n"
    s += "    *domPosY = convert_floating_t(geoDomPosTemplatePositionsY_flat[index])*GEO_DOM_POS_MAX_ABS_Y_MULTIPLIER_IN_TEMPLATE + geoDomPosStringMeanPosY[stringNum];\n"
    s += "    *domPosZ = geoDomPosTemplatePositionsZ_flat[index];\n}\n\n"
    s += "#define NUM_STRINGS %d\n#define OM_RADIUS %s\n" % (geo["num_strings"], hexf(geo["om_radius"]))
    s += "#define GEO_LAYER_STRINGSET_NUM %d\n#define GEO_LAYER_STRINGSET_MAX_NUM_LAYERS %d\n" % (geo["num_sets"], geo["max_layers"])
    s += arr("float", "geoStringPosX", "NUM_STRINGS", geo["str_x"], hexf) + arr("float", "geoStringPosY", "NUM_STRINGS", geo["str_y"], hexf)
    s += "#define GEO_STRING_MAX_RADIUS %s\n" % hexf(geo["string_max_radius"])
    s += arr("float", "geoStringRadius", "NUM_STRINGS", geo["str_radius"], hexf)
    s += arr("float", "geoStringMinZ", "NUM_STRINGS", geo["str_minz"], hexf) + arr("float", "geoStringMaxZ", "NUM_STRINGS", geo["str_maxz"], hexf)
    s += "#define GEO_CELL_NUM_SUBDETECTORS %d\n" % len(geo["cells"])
    for k, c in enumerate(geo["cells"]):
        sfx = "_%d" % k
        s += "#define GEO_CELL_NUM_X%s %d\n#define GEO_CELL_NUM_Y%s %d\n" % (sfx, c["nx"], sfx, c["ny"])
        s += "#define GEO_CELL_WIDTH_X%s %s\n#define GEO_CELL_WIDTH_Y%s %s\n" % (sfx, hexf(c["width_x"]), sfx, hexf(c["width_y"]))
        s += "#define GEO_CELL_START_X%s %s\n#define GEO_CELL_START_Y%s %s\n" % (sfx, hexf(c["start_x"]), sfx, hexf(c["start_y"]))
        s += arr("unsigned short", "geoCellIndex" + sfx, "GEO_CELL_NUM_X%s*GEO_CELL_NUM_Y%s" % (sfx, sfx), c["index"], u16)
    s += arr("unsigned char", "geoStringInStringSet", "NUM_STRINGS", geo["str_set"], ints)
    s += arr("unsigned short", "geoLayerNum", "GEO_LAYER_STRINGSET_NUM", geo["set_nlayers"], ints)
    s += arr("float", "geoLayerStartZ", "GEO_LAYER_STRINGSET_NUM", geo["set_startz"], hexf)
    s += arr("float", "geoLayerHeight", "GEO_LAYER_STRINGSET_NUM", geo["set_height"], hexf)
    s += "#define GEO_geoLayerToOMNumIndexPerStringSet_BUFFER_SIZE %d\n" % len(geo["layer_to_om"])
    return s


def emit_polynomial(name, coefficients):
    """function/I3CLSimFunctionPolynomial.cxx:104-156, no range limits (two or more coefficients)"""
    assert len(coefficients) >= 2
    body = "".join("%s + x*(" % fl(c) for c in coefficients[:-1]) + fl(coefficients[-1]) + ")" * (len(coefficients) - 1)
    return "inline float %s(float x);\ninline float %s(float x)\n{\nreturn %s;\n}\n" % (name, name, body)


def emit_binning_code(tb):
    """tabulator/Axes.cxx:68-117 (GenerateBinningCode for spherical axes), Axis.cxx:45-60, 110-113, 151-171"""
    assert tb["kind"] == "spherical"
    s = ""
    if tb["full_azimuth"]:
        s += "#define HAS_FULL_AZIMUTH_EXTENSION\n"
    with open(os.path.join(KERNELS, "spherical_coordinates.c.cl")) as f:
        s += f.read() + "\n"
    s += "inline bool isOutOfBounds(const coordinate_t coords)\n{\n    return (coords.s3 > %s)|| (coords.s0 > %s);\n}\n\n" % (
        hexf(tb["max3"]), hexf(tb["max0"]))
    terms = []
    for i, ax in enumerate(tb["axes"]):
        var = "coords.s%d" % i
        power = ax["power"] if ax["kind"] == "power" else 1
        inv = {1: var, 2: "sqrt(%s)" % var, 3: "cbrt(%s)" % var}.get(power, "pow(%s, %s)" % (var, fl(1.0 / power)))
        terms.append("%d*(clamp(convert_int_sat_rtn(%s*%s - %s), -1, %d)+1)" % (tb["strides"][i], hexf(tb["scale"][i]), inv, hexf(tb["offset"][i]), ax["n_bins"]))
    s += "inline uint getBinIndex(coordinate_t coords)\n{\n    return " + "\n         + ".join(terms) + ";\n}\n\n"
    return s


def emit_tabulate_program(medium, generators, bias, tb):
    """tabulator/I3CLSimStepToTableConverter.cxx:178-207: the table maker's program (no geometry: SAVE_ALL_PHOTONS)"""
    from oracle import builders as B

    def kernel(name):
        with open(os.path.join(KERNELS, name)) as f:
            return f.read()
    s = emit_preamble(1.0, stop_detected=False)
    s += "#define SAVE_ALL_PHOTONS\n#define SAVE_ALL_PHOTONS_PRESCALE 1\n#define PROPAGATE_FOR_FIXED_NUMBER_OF_ABSORPTION_LENGTHS 42\n#define TABULATE\n"
    if len(tb["axes"]) > 4:
        s += "#define TABULATE_IMPACT_ANGLE\n"
    s += "#define TABLE_ENTRIES_PER_STREAM %d\n#define VOLUME_MODE_STEP %s\n" % (tb["entries_per_stream"], hexf(tb["volume_step"]))
    s += "__constant floating_t min_invGroupVel = %s;\n__constant floating_t tan_thetaC = %s;\n" % (hexf(tb["min_inv_groupvel"]), hexf(tb["tan_thetac"]))
    s += kernel("mwcrng_kernel.cl")
    s += emit_wavelength_generators(generators)
    s += emit_function_from_table("getWavelengthBias", bias) + "\n"
    s += emit_medium(medium)
    s += emit_polynomial("getAngularAcceptance", tb["angular"])
    # saveHit() (c.cl:307-404, compiled but never called under TABULATE) calls geometryGetDomPosition, which only the geometry
    # source defines -- and the table maker's program has none (StepToTableConverter.cxx:196-206): at this revision the
    # reference's own table-maker program does not compile either.  An empty definition stands in for it here.
    s += ("inline void geometryGetDomPosition(unsigned short stringNum, unsigned short domNum, floating_t *domPosX, floating_t *domPosY, "
          "floating_t *domPosZ) { *domPosX = 0.f; *domPosY = 0.f; *domPosZ = 0.f; }\n")
    s += kernel("propagation_kernel.h.cl")
    s += emit_binning_code(tb)
    s += kernel("propagation_kernel.c.cl")
    return s


def emit_program(medium, geo, generators, bias, pancake, fixed_abs=None, history_n=0, stop_detected=True):
    """OpenCL.cxx:659-667: preamble, RNG, wavelength generators, bias, medium, geometry, kernels"""
    def kernel(name):
        with open(os.path.join(KERNELS, name)) as f:
            return f.read()
    s = emit_preamble(pancake, stop_detected=stop_detected, fixed_abs=fixed_abs, history_n=history_n)
    s += kernel("mwcrng_kernel.cl")
    s += emit_wavelength_generators(generators)
    s += emit_function_from_table("getWavelengthBias", bias) + "\n"
    s += emit_medium(medium)
    s += emit_geometry(geo)
    for name in ("propagation_kernel.h.cl", "sparse_collision_kernel.h.cl", "sparse_collision_kernel.c.cl", "propagation_kernel.c.cl"):
        s += kernel(name)
    return s


# ---------------------------------------------------------------------------------------------------------------------------
# build and run
# ---------------------------------------------------------------------------------------------------------------------------
def build(program_text, workdir, no_flasher, history, tabulate=False):
    cl = os.path.join(workdir, "program.cl")
    with open(cl, "w") as f:
        f.write(program_text)
    obj = os.path.join(workdir, "program.o")
    cmd = [CLANG, "-x", "cl", "-cl-std=CL1.2", "-Xclang", "-finclude-default-header", "-target", "x86_64-unknown-linux-gnu", "-O2",
           "-ffp-contract=off", "-fPIC", "-Dinline=static inline", "-Wno-everything", "-c", cl, "-o", obj]
    if no_flasher:
        cmd.insert(-4, "-DNO_FLASHER")                  # OpenCL.cxx:648-650: only the Cherenkov spectrum exists
    subprocess.check_call(cmd)
    so = os.path.join(workdir, "libverbatim.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-I" + os.path.join(ROOT, "oracle"),
                           "-DVERBATIM_HISTORY=%d" % (1 if history else 0), "-DVERBATIM_TABULATE=%d" % (1 if tabulate else 0), "-o", so, os.path.join(ROOT, "tools", "cl_shim.cpp"), obj])
    return so


def run_verbatim(so, geo, steps, x, a, capacity, history_n=0):
    lib = C.CDLL(so)
    lib.verbatim_run.restype = C.c_uint32
    lib.verbatim_run.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    n = len(steps)
    out = np.zeros((capacity, 80), dtype=np.uint8)
    hist = np.zeros((capacity, max(history_n, 1), 4), dtype=np.float32)
    xs = np.ascontiguousarray(x, dtype=np.uint64).copy()
    a32 = np.ascontiguousarray(a, dtype=np.uint32).copy()
    lto = np.ascontiguousarray(geo["layer_to_om"], dtype=np.uint16)
    st = np.ascontiguousarray(steps)
    cnt = lib.verbatim_run(out.ctypes.data, capacity, lto.ctypes.data, st.ctypes.data, xs.ctypes.data, a32.ctypes.data, hist.ctypes.data, n)
    k = min(cnt, capacity)
    return out[:k], int(cnt), xs, hist[:k]


def used_history_entries(photons, rings):
    """the ring entries that hold scatter points, in forward order (OpenCL.cxx:940-989), the rest zeroed"""
    n = rings.shape[1]
    out = np.zeros_like(rings)
    for i, ns in enumerate(photons["numScatters"]):
        ns = int(ns)
        recorded = min(ns, n)
        cur = 0 if ns <= n else ns % n
        for j in range(recorded):
            out[i, j] = rings[i, cur]
            cur = (cur + 1) % n
    return out


# name -> (configuration of tests/common.py, converter options): the static kernel files' #ifdef branches in use
CASES = {
    "c1": ("c1", {}), "mie": ("mie", {}), "lea": ("lea", {}), "flasher": ("flasher", {}),
    "photonics_mie": ("photonics_mie", {}),                         # per-layer tables in 16 bits, tabulated refractive indices, HG only
    "mie_history": ("mie", dict(history=4)),                        # SAVE_PHOTON_HISTORY
    "mie_fixed_abs": ("mie", dict(fixed_abs=1.5)),                  # PROPAGATE_FOR_FIXED_NUMBER_OF_ABSORPTION_LENGTHS
    "lea_no_pancake": ("lea", dict(pancake=1.0)),                   # no PANCAKE_FACTOR
    # without STOP_PHOTONS_ON_DETECTION (SetStopDetectedPhotons(false)): every DOM on the way is saved, the photon travels on.
    # 60 strings: the reference's per-string DOM bit mask is indexed with stringNum/64 (sparse_collision_kernel.c.cl:103-104)
    # and has (GEO_MAX_DOM_INDEX+63)/64 = 1 word here, so a 65th string makes it read and write past its array
    "c1_keep": ("c1", dict(stop_detected=False)), "mie_60_keep": ("mie_60", dict(stop_detected=False)),
    "flasher_60_keep": ("flasher_60", dict(stop_detected=False)), "clear_60_keep": ("clear_60", dict(stop_detected=False)),
    "lea_dispersion": ("lea_dispersion", {}),                       # no group refractive index override: getGroupVelocity from getDispersion
    "flasher_led405": ("flasher_led405", {}),                       # the LED's measured spectrum: InterpolatedDistribution with its own x values
    "clear_keep": ("clear", dict(stop_detected=False)),             # 86 strings: strings 64-85 index the DOM mask out of bounds (informational)
    "lea_60_keep_history": ("lea_60", dict(stop_detected=False, history=4)),
    # the table maker's kernel (-DTABULATE, spherical_coordinates.c.cl): 4 axes; azimuth to 360 degrees; impact-angle axis;
    # too little entry space
    "tabulate": ("mie", {}), "tabulate360": ("mie", {}), "tabulate5": ("mie", {}), "tabulate_overflow": ("mie", {}),
}


def check_tabulate(case, write_fixtures):
    """the TABULATE variant (propagation_kernel.c.cl:228-303, 755-785 + spherical_coordinates.c.cl) against oracle_tabulate:
    table entries (bin index, weight) of every stream in order, entry counts, photons left and RNG states"""
    from clsim_amd import synthetic as S
    from oracle import builders as B, capi
    from tests import common
    cfg = common.config("mie")
    angular = [0.32813, 0.63899, 0.20049, -1.2250, -0.14470, 4.1695, 0.76898, -5.8690, -2.0939, 2.3834, 1.0435]
    if case == "tabulate5":
        axes = [B.power_axis(0, 580, 40, 2), B.linear_axis(0, 180, 8), B.linear_axis(-1, 1, 20), B.power_axis(0, 7e3, 21, 2), B.linear_axis(-1, 1, 10)]
    elif case == "tabulate360":
        axes = [B.power_axis(0, 300, 30, 2), B.linear_axis(0, 360, 24), B.linear_axis(-1, 1, 20), B.power_axis(0, 3e3, 30, 2)]
    else:
        axes = [B.power_axis(0, 580, 40, 2), B.linear_axis(0, 180, 8), B.linear_axis(-1, 1, 20), B.power_axis(0, 7e3, 21, 2)]
    eps = 6000 if case == "tabulate_overflow" else 40000              # too few entries: streams stop early and rewind (c.cl:770-776)
    tb = B.tabulator_config("spherical", axes, cfg["med_o"], angular, entries_per_stream=eps)
    bias = B.icecube_dom_acceptance()
    gens = [B.cherenkov_wlen_generator(bias, cfg["med_o"])]
    g = cfg["geom"]
    geo = B.build_geometry(g["string_ids"], g["dom_ids"], g["x"], g["y"], g["z"], g["subdetectors"], g["om_radius"])
    T = capi.make_tables(cfg["med_o"], geo, gens, bias, pancake=1.0, tabulator=tb)
    n = 64
    steps = S.cascade_steps(n, seed=5, vertex=(3.0, -2.0, 10.0), photons_per_step=12, pad_to=n)
    x, a = common.streams(n)
    ref = B.reference_particle((1.0, 0.5, -2.0), 3.0, (0.3, -0.2, 0.9327379053))
    ent_o, num_o, left_o, x_o = capi.tabulate(T, steps, x, a, ref, threads=8)
    text = emit_tabulate_program(cfg["med_o"], gens, bias, tb)
    with tempfile.TemporaryDirectory() as d:
        so = build(text, d, no_flasher=True, history=False, tabulate=True)
        lib = C.CDLL(so)
        lib.verbatim_tabulate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        st = np.ascontiguousarray(steps).copy()
        ent_v = np.zeros((n, eps), dtype=capi.ENTRY_DTYPE)
        num_v = np.zeros(n, dtype=np.uint32)
        xs = np.ascontiguousarray(x, dtype=np.uint64).copy()
        a32 = np.ascontiguousarray(a, dtype=np.uint32).copy()
        refv = np.ascontiguousarray(ref, dtype=np.float32)
        lib.verbatim_tabulate(st.ctypes.data, refv.ctypes.data, ent_v.ctypes.data, num_v.ctypes.data, xs.ctypes.data, a32.ctypes.data, n)
    # the kernel leaves the photons it could not finish in inputSteps[i].numPhotons (c.cl:773); a finished step keeps its count
    left_v = np.where(num_v >= eps - 0, st["num"], 0).astype(np.uint32)
    left_v = np.where(st["num"] != steps["num"], st["num"], 0).astype(np.uint32)
    same_num = np.array_equal(num_v, num_o)
    same_left = np.array_equal(left_v, left_o)
    same_rng = np.array_equal(xs, x_o)
    same_entries = same_num and all(ent_v[i, :num_v[i]].tobytes() == ent_o[i, :num_o[i]].tobytes() for i in range(n))
    print("%-18s %4d streams %8d entries: entry counts %s | entries (bin, weight) %s | photons left (%d) %s | RNG states %s"
          % (case, n, int(num_v.sum()), "IDENTICAL" if same_num else "DIFFER", "IDENTICAL" if same_entries else "DIFFER", int(left_v.sum()),
             "IDENTICAL" if same_left else "DIFFER", "IDENTICAL" if same_rng else "DIFFER"), flush=True)
    ok = same_num and same_entries and same_left and same_rng
    if write_fixtures and ok:
        out = os.path.join(ROOT, "tests", "golden", "verbatim_cl_%s.npz" % case)
        # the entry stream itself is 10 MB: stored are its SHA-256 (stream after stream, entries in order), the table it
        # adds up to in binary64 (the order-independent sum) and the per-stream bookkeeping
        import hashlib
        flat = np.concatenate([ent_v[i, :num_v[i]] for i in range(n)])
        bins = np.zeros(tb["n_bins"], dtype=np.float64)
        np.add.at(bins, flat["index"], flat["weight"].astype(np.float64))
        nz = np.nonzero(bins)[0]
        if len(nz) > 100000:             # (the five-dimensional table: the hash of the entry stream says it all; keep the fixture small)
            nz = nz[:0]
        np.savez_compressed(out, num=num_v, left=left_v, rng_x=xs, entries_sha256=np.frombuffer(hashlib.sha256(flat.tobytes()).digest(), dtype=np.uint8),
                            bins_nonzero=nz.astype(np.uint32), bins_sum=bins[nz], n_bins=np.int64(tb["n_bins"]), entries_per_stream=np.int64(eps))
        print("   wrote", os.path.relpath(out, ROOT))
    return ok


def check_config(case, n_steps, write_fixtures):
    if case.startswith("tabulate"):
        return check_tabulate(case, write_fixtures)
    from clsim_amd.synthetic import PHOTON_DTYPE
    from oracle import builders as B, capi
    from tests import common
    name, opt = CASES[case]
    pancake, fixed_abs, history = opt.get("pancake", 5.0), opt.get("fixed_abs"), opt.get("history", 0)
    stop_detected = opt.get("stop_detected", True)
    cfg = common.config(name)
    g = cfg["geom"]
    geo = B.build_geometry(g["string_ids"], g["dom_ids"], g["x"], g["y"], g["z"], g["subdetectors"], g["om_radius"])
    bias = B.icecube_dom_acceptance()
    gens = common.oracle_generators(cfg, bias)
    steps = common.steps_for(cfg, n_steps, seed=3)
    x, a = common.streams(len(steps))
    T = capi.make_tables(cfg["med_o"], geo, gens, bias, pancake=pancake, stop_detected=stop_detected, fixed_abs_lengths=fixed_abs, history_entries=history)
    if history:
        ph_o, cnt_o, x_o, _, hist_o = capi.propagate(T, steps, x, a, history=True)
    else:
        ph_o, cnt_o, x_o, _ = capi.propagate(T, steps, x, a, threads=8)
    text = emit_program(cfg["med_o"], geo, gens, bias, pancake, fixed_abs=fixed_abs, history_n=history, stop_detected=stop_detected)
    with tempfile.TemporaryDirectory() as d:
        so = build(text, d, no_flasher=not cfg["flasher"], history=bool(history))
        rec, cnt_v, x_v, hist_v = run_verbatim(so, geo, steps, x, a, max(4 * cnt_o, 1024), history)
    ph_v = np.frombuffer(rec.tobytes(), dtype=PHOTON_DTYPE)
    same_count = cnt_v == cnt_o
    same_rng = np.array_equal(x_v, x_o)
    same_hits = same_count and common.sort_photons(ph_v).tobytes() == common.sort_photons(ph_o).tobytes()
    extra = ""
    if history:
        # both sides run serially, step after step: the records come out in the same order, each with its history ring
        # (ring entries a photon never wrote are uninitialised private memory in the reference, c.cl:452-455: compared are the
        # entries ConvertPhotonHistories reads, OpenCL.cxx:940-989)
        same_hist = same_count and ph_v.tobytes() == ph_o.tobytes() and \
            np.array_equal(used_history_entries(ph_v, hist_v).view(np.uint32), used_history_entries(ph_o, hist_o).view(np.uint32))
        extra = " | photon histories %s" % ("IDENTICAL" if same_hist else "DIFFER")
        same_hits = same_hits and same_hist
    print("%-14s %6d steps %9d photons: verbatim kernel %6d hits, oracle %6d hits | hit records %s | final RNG states %s%s"
          % (case, len(steps), int(steps["num"].sum()), cnt_v, cnt_o, "IDENTICAL" if same_hits else "DIFFER", "IDENTICAL" if same_rng else "DIFFER", extra), flush=True)
    if write_fixtures and same_hits and same_rng:
        out = os.path.join(ROOT, "tests", "golden", "verbatim_cl_%s.npz" % case)
        data = dict(n_steps=np.int64(n_steps), seed=np.int64(3), hits=common.sort_photons(ph_v).view(np.uint8).reshape(-1, 80), rng_x=x_v)
        if history:
            data.update(unsorted_hits=ph_v.view(np.uint8).reshape(-1, 80), histories=used_history_entries(ph_v, hist_v))
        np.savez_compressed(out, **data)
        print("   wrote", os.path.relpath(out, ROOT))
    return same_hits and same_rng


def check_refused_modes():
    """The two modes the HIP converter refuses at Compile(): does the reference's own program compile in them?  (SPICE-Mie: no
    direction transforms, the friendliest case.)  Returns {mode: first compiler error or None}."""
    from oracle import builders as B
    from tests import common
    cfg = common.config("mie")
    g = cfg["geom"]
    geo = B.build_geometry(g["string_ids"], g["dom_ids"], g["x"], g["y"], g["z"], g["subdetectors"], g["om_radius"])
    bias = B.icecube_dom_acceptance()
    gens = common.oracle_generators(cfg, bias)
    single = ("typedef float floating_t;\ntypedef float2 floating2_t;\ntypedef float4 floating4_t;\n"
              "#define convert_floating_t convert_float\n#define ZERO 0.f\n#define ONE 1.f\n")
    # I3CLSimHelperMath.cxx:16-25
    double = ("#pragma OPENCL EXTENSION cl_khr_fp64 : enable\ntypedef double floating_t;\ntypedef double2 floating2_t;\n"
              "typedef double4 floating4_t;\n#define convert_floating_t convert_double\n#define DOUBLE_PRECISION\n#define ZERO 0.\n#define ONE 1.\n")
    programs = {}
    text = emit_program(cfg["med_o"], geo, gens, bias, 5.0)
    assert single in text
    programs["DOUBLE_PRECISION"] = text.replace(single, double)
    # OpenCL.cxx:395-412 (no STOP_PHOTONS_ON_DETECTION, SAVE_ALL_PHOTONS + prescale), :459-470 and :655-657 (no geometry source),
    # :522-527 (no collision detection sources)
    text = emit_program(cfg["med_o"], geo, gens, bias, 5.0, stop_detected=False)
    geometry = emit_geometry(geo)
    assert geometry in text
    text = text.replace(geometry, "").replace(single, single + "#define SAVE_ALL_PHOTONS\n#define SAVE_ALL_PHOTONS_PRESCALE 0.01f\n")
    for name in ("sparse_collision_kernel.h.cl", "sparse_collision_kernel.c.cl"):
        with open(os.path.join(KERNELS, name)) as f:
            part = f.read()
        assert part in text
        text = text.replace(part, "")
    programs["SAVE_ALL_PHOTONS"] = text
    out = {}
    for mode, program in programs.items():
        with tempfile.TemporaryDirectory() as d:
            cl = os.path.join(d, "program.cl")
            with open(cl, "w") as f:
                f.write(program)
            r = subprocess.run([CLANG, "-x", "cl", "-cl-std=CL1.2", "-Xclang", "-finclude-default-header", "-target", "x86_64-unknown-linux-gnu", "-O2",
                                "-ffp-contract=off", "-fPIC", "-Dinline=static inline", "-DNO_FLASHER", "-c", cl, "-o", os.path.join(d, "p.o")],
                               capture_output=True, text=True)
            errors = [l.split("error: ", 1)[1] for l in r.stderr.splitlines() if "error: " in l]
            out[mode] = (errors[0] if errors else "compiler failed") if r.returncode != 0 else None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default=",".join(CASES))
    ap.add_argument("--steps", type=int, default=4096)
    ap.add_argument("--write-fixtures", action="store_true")
    ap.add_argument("--refused-modes", action="store_true", help="try to compile the reference's program with DOUBLE_PRECISION and with SAVE_ALL_PHOTONS")
    args = ap.parse_args()
    if not os.path.isdir(KERNELS):
        raise SystemExit("the reference tree is not on this machine: this check runs in the build container only")
    if args.refused_modes:
        for mode, error in check_refused_modes().items():
            print("%-18s %s" % (mode, "compiles" if error is None else "DOES NOT COMPILE: " + error))
        return
    ok = True
    for name in args.configs.split(","):
        ok = check_config(name, 1000 if name == "c1" else (2048 if name in ("mie_history", "photonics_mie") else args.steps), args.write_fixtures) and ok
    raise SystemExit(0 if ok else 1)


if __name__ == "__main__":
    main()
