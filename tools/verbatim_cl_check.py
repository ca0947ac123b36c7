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

usage: tools/verbatim_cl_check.py [--configs c1,mie,lea,flasher] [--steps 4096] [--write-fixtures]
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
def emit_preamble(pancake, stop_detected=True):
    """I3CLSimHelperMath.cxx:16-46 (single precision) + OpenCL.cxx:390-442 (mode #defines)"""
    s = ("#pragma OPENCL EXTENSION cl_khr_global_int32_base_atomics : enable\n"
         "#pragma OPENCL EXTENSION cl_khr_byte_addressable_store : enable\n"
         "typedef float floating_t;\ntypedef float2 floating2_t;\ntypedef float4 floating4_t;\n"
         "#define convert_floating_t convert_float\n#define ZERO 0.f\n#define ONE 1.f\n\n")
    if stop_detected:
        s += "#define STOP_PHOTONS_ON_DETECTION\n"
    if pancake != 1.0:
        s += "#define PANCAKE_FACTOR %s\n" % fl(pancake)
    return s


def emit_interpolated_distribution(name, gen):
    """random_value/I3CLSimRandomValueInterpolatedDistribution.cxx:177-336, constant spacing"""
    from oracle import builders as B
    yv, ycum = B.interp_dist_tables(gen)
    n = len(yv)
    p = "_" + name
    s = "#define %sNUM_DIST_ENTRIES %d\n" % (p, n)
    s += "__constant float %sdistYValues[%sNUM_DIST_ENTRIES] = {%s};\n" % (p, p, ", ".join(hexf(v) for v in yv))
    s += "__constant float %sdistYCumulativeValues[%sNUM_DIST_ENTRIES] = {%s};\n" % (p, p, ", ".join(hexf(v) for v in ycum))
    sp, first = fl(gen["spacing"]), fl(gen["first"])
    s += "inline float %s(RNG_ARGS);\ninline float %s(RNG_ARGS)\n{\n" % (name, name)
    s += "    const float randomNumber = RNG_CALL_UNIFORM_OC;\n"
    s += "    unsigned int k=0;\n    float this_acu = 0.f;\n"
    s += "    for (;;)\n    {\n        float next_acu = %sdistYCumulativeValues[k+1];\n" % p
    s += "        if (next_acu >= randomNumber) break;\n        this_acu = next_acu;\n        ++k;\n    }\n"
    s += "    const float b = %sdistYValues[k];\n" % p
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
        if g["kind"] == "interp":
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
    assert "phase_table" not in m and "group_table" not in m, "tabulated refractive indices: not emitted by this tool"
    s += emit_refindex("getPhaseRefIndex_func0", m["n"], m["g"], "phase") + emit_single_layer_wrapper("getPhaseRefIndex")
    s += emit_refindex("getGroupRefIndex_func0", m["n"], m["g"], "group") + emit_single_layer_wrapper("getGroupRefIndex")
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


def emit_program(medium, geo, generators, bias, pancake):
    """OpenCL.cxx:659-667: preamble, RNG, wavelength generators, bias, medium, geometry, kernels"""
    def kernel(name):
        with open(os.path.join(KERNELS, name)) as f:
            return f.read()
    s = emit_preamble(pancake)
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
def build(program_text, workdir, no_flasher):
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
                           "-o", so, os.path.join(ROOT, "tools", "cl_shim.cpp"), obj])
    return so


def run_verbatim(so, geo, steps, x, a, capacity):
    lib = C.CDLL(so)
    lib.verbatim_run.restype = C.c_uint32
    lib.verbatim_run.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    n = len(steps)
    out = np.zeros((capacity, 80), dtype=np.uint8)
    xs = np.ascontiguousarray(x, dtype=np.uint64).copy()
    a32 = np.ascontiguousarray(a, dtype=np.uint32).copy()
    lto = np.ascontiguousarray(geo["layer_to_om"], dtype=np.uint16)
    st = np.ascontiguousarray(steps)
    cnt = lib.verbatim_run(out.ctypes.data, capacity, lto.ctypes.data, st.ctypes.data, xs.ctypes.data, a32.ctypes.data, None, n)
    return out[:min(cnt, capacity)], int(cnt), xs


def check_config(name, n_steps, write_fixtures):
    from clsim_amd.synthetic import PHOTON_DTYPE
    from oracle import builders as B, capi
    from tests import common
    cfg = common.config(name)
    g = cfg["geom"]
    geo = B.build_geometry(g["string_ids"], g["dom_ids"], g["x"], g["y"], g["z"], g["subdetectors"], g["om_radius"])
    bias = B.icecube_dom_acceptance()
    gens = [B.cherenkov_wlen_generator(bias, cfg["med_o"])]
    if cfg["flasher"]:
        gens.append(dict(kind="const", value=common.FLASHER_WLEN))
    steps = common.steps_for(cfg, n_steps, seed=3)
    x, a = common.streams(len(steps))
    T = capi.make_tables(cfg["med_o"], geo, gens, bias, pancake=5.0)
    ph_o, cnt_o, x_o, _ = capi.propagate(T, steps, x, a, threads=8)
    text = emit_program(cfg["med_o"], geo, gens, bias, 5.0)
    with tempfile.TemporaryDirectory() as d:
        so = build(text, d, no_flasher=not cfg["flasher"])
        rec, cnt_v, x_v = run_verbatim(so, geo, steps, x, a, max(4 * cnt_o, 1024))
    ph_v = np.frombuffer(rec.tobytes(), dtype=PHOTON_DTYPE)
    same_count = cnt_v == cnt_o
    same_rng = np.array_equal(x_v, x_o)
    same_hits = same_count and common.sort_photons(ph_v).tobytes() == common.sort_photons(ph_o).tobytes()
    print("%-8s %6d steps %9d photons: verbatim kernel %6d hits, oracle %6d hits | hit records %s | final RNG states %s"
          % (name, len(steps), int(steps["num"].sum()), cnt_v, cnt_o, "IDENTICAL" if same_hits else "DIFFER", "IDENTICAL" if same_rng else "DIFFER"), flush=True)
    if write_fixtures and same_hits and same_rng:
        out = os.path.join(ROOT, "tests", "golden", "verbatim_cl_%s.npz" % name)
        np.savez_compressed(out, n_steps=np.int64(n_steps), seed=np.int64(3), hits=common.sort_photons(ph_v).view(np.uint8).reshape(-1, 80),
                            rng_x=x_v)
        print("   wrote", os.path.relpath(out, ROOT))
    return same_hits and same_rng


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="c1,mie,lea,flasher")
    ap.add_argument("--steps", type=int, default=4096)
    ap.add_argument("--write-fixtures", action="store_true")
    args = ap.parse_args()
    if not os.path.isdir(KERNELS):
        raise SystemExit("the reference tree is not on this machine: this check runs in the build container only")
    ok = True
    for name in args.configs.split(","):
        ok = check_config(name, 1000 if name == "c1" else args.steps, args.write_fixtures) and ok
    raise SystemExit(0 if ok else 1)


if __name__ == "__main__":
    main()
