/*
 * uam_pmath.h — portable fp64 sin/cos/log built only from IEEE-754 basic
 * operations (+ - * / sqrt fma), so that the SAME source gives bit-identical
 * results on the host (gcc) and on gfx950 (hipcc) when both are compiled with
 * -ffp-contract=off.  The env kernels use these instead of libm/ocml wherever
 * the reference calls math.cos / math.sin (Simulators/SingleAircraftEnv.py:275-276,
 * 306-307) or NumPy's rk_gauss log (legacy-distributions), which is what makes
 * the device path bit-reproducible against the CPU oracle.
 *
 * Accuracy (measured in tests/test_pmath.py against libm): < 1 ulp for
 * |x| < 2^20 (sincos) and for x in (0, 1] (log).  Not a general libm: no
 * inf/nan/subnormal handling beyond what the env can produce.
 */
#ifndef UAM_PMATH_H
#define UAM_PMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define UAM_HD __host__ __device__ __forceinline__
#else
#define UAM_HD static inline
#endif

UAM_HD double uam_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

UAM_HD uint64_t uam_d2u(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
UAM_HD double   uam_u2d(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }

#define UAM_PM_NAME(f) f
#define UAM_KC(c) (c)
#define UAM_FMAK(a, z, c) uam_fma(a, z, c)
#include "uam_pmath_impl.h"
#undef UAM_PM_NAME
#undef UAM_KC
#undef UAM_FMAK

/* Second flavour for code that calls these inside long loops (the MCTS kernel): every coefficient is
 * pinned to a scalar register pair at its point of use (an opaque, non-hoistable s_mov pair).  Without
 * it the compiler hoists all ~60 fp64 literals into VGPRs for the whole kernel (197 VGPRs, 2 waves per
 * SIMD).  Same operations on the same values: results are bit-identical to the plain flavour. */
#if defined(__HIPCC__)
__host__ __device__ __forceinline__ double uam_kc_dev(double v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    unsigned int lo = (unsigned int)u, hi = (unsigned int)(u >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
#else
    return v;
#endif
}
/* A Horner step of that flavour, fma(a, z, c) with the coefficient as the SCALAR addend of one v_fma_f64.  Left to
 * the compiler, a pinned coefficient is first copied into a VGPR pair (two v_mov) to serve as the tied addend of
 * v_fmac_f64: three vector instructions per step instead of one, ~40 of a sincos' ~110.  Same IEEE fma. */
__host__ __device__ __forceinline__ double uam_fmak_dev(double a, double z, double c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(z), "s"(c));
    return d;
#else
    return __builtin_fma(a, z, c);
#endif
}
#define UAM_PM_NAME(f) f##_k
#define UAM_KC(c) uam_kc_dev(c)
#define UAM_FMAK(a, z, c) uam_fmak_dev(a, z, uam_kc_dev(c))
#include "uam_pmath_impl.h"
#undef UAM_PM_NAME
#undef UAM_KC
#undef UAM_FMAK
#endif

/* sin and cos of the angle 2*pi*w/2^32 in float32, built from the integer: the top two bits pick
 * the quadrant exactly, 24 further bits give the position t in [0,1) inside it, and the octant is
 * folded so the Taylor kernels only see x in [0, pi/4].  |error| < 1.5e-7.  Used for the heading of
 * freshly spawned intruders in the float32 Philox mode (explicit fmaf: bit-identical host/device). */
UAM_HD void uam_sincos_turn_f32(uint32_t w, float* sn, float* cs)
{
    const uint32_t quad = w >> 30;
    float t = (float)((w >> 6) & 0xFFFFFFu) * 0x1p-24f;
    const int fold = t > 0.5f;
    if (fold) t = 1.0f - t;
    const float x = 0x1.921fb6p+0f * t;
    const float z = x * x;
    float ps = 0x1.71de3ap-19f;
    ps = __builtin_fmaf(ps, z, -0x1.a01a02p-13f);
    ps = __builtin_fmaf(ps, z, 0x1.111112p-7f);
    ps = __builtin_fmaf(ps, z, -0x1.555556p-3f);
    const float s = __builtin_fmaf(x * z, ps, x);
    float pc = -0x1.27e4fcp-22f;
    pc = __builtin_fmaf(pc, z, 0x1.a01a02p-16f);
    pc = __builtin_fmaf(pc, z, -0x1.6c16c2p-10f);
    pc = __builtin_fmaf(pc, z, 0x1.555556p-5f);
    pc = __builtin_fmaf(pc, z, -0.5f);
    const float c = __builtin_fmaf(z, pc, 1.0f);
    const float qs = fold ? c : s, qc = fold ? s : c;   /* sin, cos of the in-quadrant angle */
    float rs = (quad & 1u) ? qc : qs;
    float rc = (quad & 1u) ? qs : qc;
    if (quad == 1u || quad == 2u) rc = -rc;
    if (quad >= 2u) rs = -rs;
    *sn = rs;
    *cs = rc;
}

/* float32 sin and cos of a float64 heading (any magnitude): the heading is reduced to a fraction of a turn in
 * float64 (multiply, floor, subtract: exact IEEE operations, identical on host and device) and the top 26 bits of
 * that fraction go through uam_sincos_turn_f32.  Angular resolution 2*pi/2^26 = 9.4e-8 rad.  Used for the
 * float32 velocities derived from held headings in the float32 Philox mode (RandIntruder heading changes,
 * spawns of envs that hold speed/heading): keeps fp64 trig out of those sweep instances. */
UAM_HD void uam_sincos_heading_f32(double heading, float* sn, float* cs)
{
    double t = heading * 0x1.45f306dc9c883p-3;      /* 1 / (2 pi) */
    t = t - __builtin_floor(t);
    const double s = t * 4294967296.0;
    const uint32_t w = s >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)s;
    uam_sincos_turn_f32(w, sn, cs);
}

#endif /* UAM_PMATH_H */
