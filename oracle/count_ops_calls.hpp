/*
 * count_ops_calls.hpp -- TEST / MEASUREMENT INFRASTRUCTURE (counting build of the oracle, see count_ops.hpp).
 * The math library by name: every call of oracle_math.h that clsim_oracle.c makes is one counted unit of that name; the
 * library itself stays plain float, so its inner operations are not counted as reference arithmetic.
 */
#pragma once
static inline cfloat oc_log(cfloat x) { oc_count(OC_LOG); return cfloat(om_log(x.v)); }
static inline cfloat oc_exp(cfloat x) { oc_count(OC_EXP); return cfloat(om_exp(x.v)); }
static inline cfloat oc_powr(cfloat x, cfloat y) { oc_count(OC_POWR); return cfloat(om_powr(x.v, y.v)); }
static inline cfloat oc_powr_unit(cfloat x, cfloat y) { oc_count(OC_POWR_UNIT); return cfloat(om_powr_unit(x.v, y.v)); }
static inline cfloat oc_pow_frac(cfloat x, cfloat y) { oc_count(OC_OTHER_MATH); return cfloat(om_pow_frac(x.v, y.v)); }
static inline cfloat oc_cbrt(cfloat x) { oc_count(OC_OTHER_MATH); return cfloat(om_cbrt(x.v)); }
static inline cfloat oc_sqrt(cfloat x) { oc_count(OC_SQRT); return cfloat(om_sqrt(x.v)); }
static inline cfloat oc_rsqrt(cfloat x) { oc_count(OC_RSQRT); return cfloat(om_rsqrt(x.v)); }
static inline cfloat oc_fabs(cfloat x) { oc_count(OC_FABS); return cfloat(om_fabs(x.v)); }
static inline cfloat oc_sin(cfloat x) { oc_count(OC_SIN); return cfloat(om_sin(x.v)); }
static inline cfloat oc_cos(cfloat x) { oc_count(OC_COS); return cfloat(om_cos(x.v)); }
static inline void oc_sincos(cfloat x, cfloat *s, cfloat *c) { oc_count(OC_SINCOS); float a, b; om_sincos(x.v, &a, &b); s->v = a; c->v = b; }
static inline cfloat oc_acos(cfloat x) { oc_count(OC_ACOS); return cfloat(om_acos(x.v)); }
static inline cfloat oc_acos_f(cfloat x) { oc_count(OC_ACOS); return cfloat(om_acos_f(x.v)); }
static inline cfloat oc_atan2(cfloat y, cfloat x) { oc_count(OC_ATAN2); return cfloat(om_atan2(y.v, x.v)); }
static inline cfloat oc_truncf(cfloat x) { oc_count(OC_FLOOR_TRUNC); return cfloat(__builtin_truncf(x.v)); }
static inline cfloat oc_floorf(cfloat x) { oc_count(OC_FLOOR_TRUNC); return cfloat(__builtin_floorf(x.v)); }
#define om_log oc_log
#define om_exp oc_exp
#define om_powr oc_powr
#define om_powr_unit oc_powr_unit
#define om_pow_frac oc_pow_frac
#define om_cbrt oc_cbrt
#define om_sqrt oc_sqrt
#define om_rsqrt oc_rsqrt
#define om_fabs oc_fabs
#define om_sin oc_sin
#define om_cos oc_cos
#define om_sincos oc_sincos
#define om_acos oc_acos
#define om_acos_f oc_acos_f
#define om_atan2 oc_atan2
#define __builtin_truncf oc_truncf
#define __builtin_floorf oc_floorf
