#!/usr/bin/env python3
"""Generates the constants of the table-driven log / sincos definitions (round 5): clsim_amd/csrc/math_tables.h (product) and
oracle/math_tables.h (checker) -- the SAME text under two include guards, because the product may not include anything from
oracle/ and the oracle must not depend on the product.  tests/test_oracle.py compares the two files' bodies.

Every value is computed with mpmath at 200 bits and printed as a C99 hexadecimal float, so the two sides (and any compiler)
read identical bit patterns.  What the constants mean is documented where they are used (oracle/oracle_math.h: om_log, om_sincos_2pi;
clsim_amd/csrc/detmath.hip.h: log_, sincos_2pi_).  usage: tools/make_math_tables.py  (build container; needs mpmath)"""
import os
import sys
import struct

import mpmath as mp

mp.mp.prec = 200
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def f32(v):
    """nearest binary32 (ties to even) of an mpmath / python number, as a python float"""
    v = mp.mpf(v)
    if abs(v) < mp.mpf(2) ** -126:          # (nothing here is meant to be subnormal: sin(pi) at 200 bits)
        return 0.0
    m, e = mp.frexp(abs(v))                 # abs(v) = m * 2^e, 0.5 <= m < 1
    q = m * mp.mpf(2) ** 24                 # 24 significant bits
    n = int(mp.nint(q))                     # mpmath nint: ties to even
    r = mp.mpf(n) * mp.mpf(2) ** (e - 24)
    out = float(r) if v > 0 else -float(r)
    assert struct.unpack("<f", struct.pack("<f", out))[0] == out
    return out


def hexf(v):
    return float(v).hex() + "f"


LN2_HI = 355.0 / 512.0                      # 9 significant bits: e * LN2_HI is exact (oracle_math.h: OM_LN2_HI)
LN2_LO = f32(mp.log(2) - mp.mpf(LN2_HI))


def log_table():
    """32 intervals of the significand m in [1, 2) (index = its top five fraction bits).  Interval j: centre c_j, INV_j = RN(1/c_j)
    (c_0 = 1 and c_31 = 2 exactly: next to one the logarithm is log1p of an EXACT r), r = fma(m, INV_j, -1);
    log m = log1p(r) - log(INV_j) with -log(INV_j) = hi_j + lo_j, hi_j a multiple of 2^-12.  The exponent arrives as
    frexp's (e + 1), so one LN2 is folded out of the constants: H_j = hi_j - LN2_HI (exact), L_j = RN(lo_j - LN2_LO)."""
    rows = []
    for j in range(32):
        if j == 0:
            inv = 1.0
        elif j == 31:
            inv = 0.5
        else:
            inv = f32(1 / (1 + (mp.mpf(j) + mp.mpf(1) / 2) / 32))
        logc = -mp.log(mp.mpf(inv))
        hi = mp.nint(logc * 4096) / 4096
        if j == 31:
            hi = mp.mpf(LN2_HI)
        lo = logc - hi
        H = hi - mp.mpf(LN2_HI)
        assert f32(H) == float(H)
        L = f32(lo - mp.mpf(LN2_LO)) if j != 31 else 0.0
        rows.append((inv, float(H), L))
    assert rows[31][1] == 0.0 and rows[31][2] == 0.0 and rows[0][1] == -LN2_HI
    return rows


def remez_like(fn, lo, hi, degree, weight=None, points=400):
    """near-minimax polynomial by iterated weighted least squares at Chebyshev nodes (good to a few per cent of the optimum,
    which is all the error budget needs); returns coefficients c0..c_degree as mpf"""
    import numpy as np
    xs = [mp.mpf(lo) + (mp.mpf(hi) - mp.mpf(lo)) * (1 - mp.cos(mp.pi * (2 * k + 1) / (2 * points))) / 2 for k in range(points)]
    ys = [fn(x) for x in xs]
    w = [mp.mpf(1) if weight is None else weight(x) for x in xs]
    A = mp.matrix(points, degree + 1)
    b = mp.matrix(points, 1)
    for i, x in enumerate(xs):
        for d in range(degree + 1):
            A[i, d] = w[i] * x ** d
        b[i] = w[i] * ys[i]
    c = mp.lu_solve(A.T * A, A.T * b)
    for _ in range(12):                     # Lawson iterations towards the minimax solution
        err = [abs(w[i] * (sum(c[d] * xs[i] ** d for d in range(degree + 1)) - ys[i])) for i in range(points)]
        top = max(err)
        if top == 0:
            break
        lw = [max(e / top, mp.mpf("1e-3")) for e in err]
        A2 = mp.matrix(points, degree + 1)
        b2 = mp.matrix(points, 1)
        for i, x in enumerate(xs):
            s = mp.sqrt(lw[i]) * w[i]
            for d in range(degree + 1):
                A2[i, d] = s * x ** d
            b2[i] = s * ys[i]
        c = mp.lu_solve(A2.T * A2, A2.T * b2)
    return [c[d] for d in range(degree + 1)]


def main():
    rows = log_table()
    # log1p(r) = r + r^2 * P(r), r in [-1/64, 1/32]: P of degree 3 with P(0) = -1/2 kept exact (an inline constant on the device)
    def lp(r):
        return (mp.log1p(r) - r) / (r * r) if r != 0 else mp.mpf(-1) / 2
    # fit (P(r) + 1/2) / r so that the constant term stays -1/2
    c = remez_like(lambda r: (lp(r) + mp.mpf(1) / 2) / r if r != 0 else mp.mpf(1) / 3, -mp.mpf(1) / 64 - mp.mpf(1) / 4096, mp.mpf(1) / 32 + mp.mpf(1) / 4096, 2)
    LOG_P = [f32(c[0]), f32(c[1]), f32(c[2])]           # r^3, r^4, r^5 coefficients (about 1/3, -1/4, 1/5)

    # sincos on [0, 2 pi]: k = rint(x * 16/pi) in 0..32, r = x - k * pi/16 (two constants), |r| <= pi/32 (+ rounding)
    H1 = f32(mp.pi / 16)
    H2 = f32(mp.pi / 16 - mp.mpf(H1))
    rmax = mp.pi / 32 * mp.mpf("1.01")
    # sin r = r + r * z * (S0 + S1 z), z = r^2
    s = remez_like(lambda z: (mp.sin(mp.sqrt(z)) - mp.sqrt(z)) / (mp.sqrt(z) * z) if z != 0 else mp.mpf(-1) / 6, 0, rmax * rmax, 1)
    # cos r - 1 = z * (-1/2 + C1 z)   (constant term kept exact)
    cc = remez_like(lambda z: ((mp.cos(mp.sqrt(z)) - 1) / z + mp.mpf(1) / 2) / z if z != 0 else mp.mpf(1) / 24, 0, rmax * rmax, 0)
    SIN_P = [f32(s[0]), f32(s[1])]
    COS_P = [f32(cc[0])]
    # The table holds sin and cos at the points the two-constant reduction actually subtracts, k * (H1 + H2), not at k pi/16:
    # x = k (H1 + H2) + r holds exactly, so sin x = S_k cos r + C_k sin r is an identity and nothing is lost next to the zeros
    # (S_16 = sin(16 (H1 + H2)) = 5.6e-15, not 0: it IS the third reduction constant's work, for free).
    sc = []
    for k in range(33):
        a = (mp.mpf(H1) + mp.mpf(H2)) * k
        sc.append((f32(mp.sin(a)), f32(mp.cos(a))))
    assert sc[0] == (0.0, 1.0)
    body = []
    body.append("/* GENERATED by tools/make_math_tables.py -- do not edit.  Constants of the table-driven log and sincos (round 5). */")
    body.append("#define MT_LN2_HI %s" % hexf(LN2_HI))
    body.append("#define MT_LN2_LO %s" % hexf(LN2_LO))
    body.append("#define MT_LOG_P3 %s" % hexf(LOG_P[0]))
    body.append("#define MT_LOG_P4 %s" % hexf(LOG_P[1]))
    body.append("#define MT_LOG_P5 %s" % hexf(LOG_P[2]))
    body.append("#define MT_SC_16OPI %s" % hexf(f32(16 / mp.pi)))
    body.append("#define MT_SC_H1 %s" % hexf(H1))
    body.append("#define MT_SC_H2 %s" % hexf(H2))
    body.append("#define MT_SIN_S0 %s" % hexf(SIN_P[0]))
    body.append("#define MT_SIN_S1 %s" % hexf(SIN_P[1]))
    body.append("#define MT_COS_C1 %s" % hexf(COS_P[0]))
    body.append("/* log: 32 rows {INV_j, H_j, L_j, 0} */")
    body.append("#define MT_LOG_ROWS 32")
    body.append("#define MT_LOG_TABLE { \\")
    for inv, H, L in rows:
        body.append("    %s, %s, %s, 0.0f, \\" % (hexf(inv), hexf(H), hexf(L)))
    body.append("}")
    body.append("/* sincos: 33 rows {sin, cos} of k * (MT_SC_H1 + MT_SC_H2) */")
    body.append("#define MT_SC_ROWS 33")
    body.append("#define MT_SC_TABLE { \\")
    for S, C in sc:
        body.append("    %s, %s, \\" % (hexf(S), hexf(C)))
    body.append("}")
    text = "\n".join(body) + "\n"
    # --check: compare with the checked-in headers and write nothing (what tests/test_oracle.py runs: a test must not touch the tree --
    # the headers are in both Makefiles' dependencies, a rewrite would rebuild every kernel); --output-dir DIR: write the two
    # headers there instead of into the tree
    check = "--check" in sys.argv[1:]
    out_dir = sys.argv[sys.argv.index("--output-dir") + 1] if "--output-dir" in sys.argv[1:] else None
    stale = []
    for path, guard in ((os.path.join(ROOT, "clsim_amd", "csrc", "math_tables.h"), "CLSIMHIP_MATH_TABLES_H"),
                        (os.path.join(ROOT, "oracle", "math_tables.h"), "CLSIM_ORACLE_MATH_TABLES_H")):
        content = "#ifndef %s\n#define %s\n%s#endif\n" % (guard, guard, text)
        if check:
            with open(path) as f:
                if f.read() != content:
                    stale.append(path)
            continue
        if out_dir is not None:
            path = os.path.join(out_dir, os.path.basename(os.path.dirname(path)) + "_" + os.path.basename(path))
        with open(path, "w") as f:
            f.write(content)
        print("wrote", path)
    if check:
        if stale:
            raise SystemExit("stale (run tools/make_math_tables.py): " + ", ".join(stale))
        print("the checked-in math tables are what the generator writes")
        return
    print("LOG_P", LOG_P, "SIN_P", SIN_P, "COS_P", COS_P, "H1", H1, "H2", H2)


if __name__ == "__main__":
    main()
