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

/* UAM_KC(c): a polynomial coefficient.  On the device it is pinned to a scalar register pair at its point
 * of use (an opaque, non-hoistable s_mov pair): inside the MCTS kernel's loops the compiler otherwise
 * hoists all ~60 fp64 literals of sincos/log into VGPRs for the whole kernel (197 VGPRs, 2 waves/SIMD).
 * Same value on both sides, so host and device results stay bit-identical. */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double uam_kc_dev(double v)
{
    unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    unsigned int lo = (unsigned int)u, hi = (unsigned int)(u >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
#define UAM_KC(c) uam_kc_dev(c)
#else
#define UAM_KC(c) (c)
#endif

UAM_HD uint64_t uam_d2u(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
UAM_HD double   uam_u2d(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }

/* sin and cos of x (|x| < ~1e6 for full accuracy).
 * Reduction: k = rint(x*2/pi), y = x - k*(C1+C2+C3) carried as y0+y1 with
 * fma-exact products; kernels are Taylor polynomials on [-pi/4, pi/4]
 * evaluated in the fdlibm "x + x^3*P" / "1 - z/2 + z^2*Q" forms. */
UAM_HD void uam_sincos(double x, double* sn, double* cs)
{
    const double INV_PIO2 = UAM_KC(0x1.45f306dc9c883p-1);
    const double C1 = UAM_KC(0x1.921fb54442d18p+0), C2 = UAM_KC(0x1.1a62633145c07p-54), C3 = UAM_KC(-0x1.f1976b7ed8fbcp-110);
    double k = __builtin_rint(x * INV_PIO2);
    double p1 = k * C1, e1 = uam_fma(k, C1, -p1);
    double p2 = k * C2, e2 = uam_fma(k, C2, -p2);
    double p3 = k * C3;
    double a = x - p1;                       /* exact (Sterbenz) */
    double s = e1 + p2;
    double bb = s - e1;
    double es = (e1 - (s - bb)) + (p2 - bb); /* two-sum error of e1+p2 */
    double lo_small = (es + e2) + p3;
    double yh = a - s;
    bb = yh - a;
    double eh = (a - (yh - bb)) + (-s - bb); /* two-sum error of a-s */
    double yl = eh - lo_small;
    double y0 = yh + yl;
    double y1 = yl - (y0 - yh);

    double z = y0 * y0;
    /* sin kernel */
    double ps = UAM_KC(0x1.71b8ef6dcf572p-66);
    ps = uam_fma(ps, z, UAM_KC(-0x1.2f49b46814157p-57));
    ps = uam_fma(ps, z, UAM_KC(0x1.952c77030ad4ap-49));
    ps = uam_fma(ps, z, UAM_KC(-0x1.ae7f3e733b81fp-41));
    ps = uam_fma(ps, z, UAM_KC(0x1.6124613a86d09p-33));
    ps = uam_fma(ps, z, UAM_KC(-0x1.ae64567f544e4p-26));
    ps = uam_fma(ps, z, UAM_KC(0x1.71de3a556c734p-19));
    ps = uam_fma(ps, z, UAM_KC(-0x1.a01a01a01a01ap-13));
    ps = uam_fma(ps, z, UAM_KC(0x1.1111111111111p-7));   /* ps = S5 + S7 z + ... */
    double v = z * y0;
    /* sin(y0+y1) = y0 + (v*(S3 + z*ps) + y1*(1 - z/2)) */
    double S = y0 + (uam_fma(v, uam_fma(z, ps, UAM_KC(-0x1.5555555555555p-3)), 0.0) + y1 * (1.0 - 0.5 * z));
    /* cos kernel */
    double pc = UAM_KC(-0x1.0ce396db7f853p-70);
    pc = uam_fma(pc, z, UAM_KC(0x1.e542ba4020225p-62));
    pc = uam_fma(pc, z, UAM_KC(-0x1.6827863b97d97p-53));
    pc = uam_fma(pc, z, UAM_KC(0x1.ae7f3e733b81fp-45));
    pc = uam_fma(pc, z, UAM_KC(-0x1.93974a8c07c9dp-37));
    pc = uam_fma(pc, z, UAM_KC(0x1.1eed8eff8d898p-29));
    pc = uam_fma(pc, z, UAM_KC(-0x1.27e4fb7789f5cp-22));
    pc = uam_fma(pc, z, UAM_KC(0x1.a01a01a01a01ap-16));
    pc = uam_fma(pc, z, UAM_KC(-0x1.6c16c16c16c17p-10));
    pc = uam_fma(pc, z, UAM_KC(0x1.5555555555555p-5));   /* pc = C4 + C6 z + ... */
    double hz = 0.5 * z;
    double w = 1.0 - hz;
    double C = w + (((1.0 - w) - hz) + (z * (z * pc) - y0 * y1));

    int64_t q = (int64_t)k;
    int n = (int)(q & 3);
    double rs = (n & 1) ? C : S;
    double rc = (n & 1) ? S : C;
    if (n == 1 || n == 2) rc = -rc;
    if (n >= 2) rs = -rs;
    *sn = rs;
    *cs = rc;
}

/* natural log for finite x > 0 (normal doubles). */
UAM_HD double uam_log(double x)
{
    const double LN2_HI = UAM_KC(0x1.62e42fee00000p-1), LN2_LO = UAM_KC(0x1.a39ef35793c76p-33);
    uint64_t u = uam_d2u(x);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    double m = uam_u2d((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL); /* [1,2) */
    if (m > UAM_KC(0x1.6a09e667f3bcdp+0)) { m *= 0.5; e += 1; }                     /* [sqrt(.5), sqrt(2)] */
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    /* log(m) = 2s + s*z*(2/3 + 2/5 z + ...) */
    double r = UAM_KC(0x1.47ae147ae147bp-4);
    r = uam_fma(r, z, UAM_KC(0x1.642c8590b2164p-4));
    r = uam_fma(r, z, UAM_KC(0x1.8618618618618p-4));
    r = uam_fma(r, z, UAM_KC(0x1.af286bca1af28p-4));
    r = uam_fma(r, z, UAM_KC(0x1.e1e1e1e1e1e1ep-4));
    r = uam_fma(r, z, UAM_KC(0x1.1111111111111p-3));
    r = uam_fma(r, z, UAM_KC(0x1.3b13b13b13b14p-3));
    r = uam_fma(r, z, UAM_KC(0x1.745d1745d1746p-3));
    r = uam_fma(r, z, UAM_KC(0x1.c71c71c71c71cp-3));
    r = uam_fma(r, z, UAM_KC(0x1.2492492492492p-2));
    r = uam_fma(r, z, UAM_KC(0x1.999999999999ap-2));
    r = uam_fma(r, z, UAM_KC(0x1.5555555555555p-1));
    /* f - s*(f - z*r): fdlibm form, exact-ish head f */
    double hfsq = 0.5 * f * f;
    double ke = (double)e;
    double t = s * (hfsq + z * r);
    return ke * LN2_HI - ((hfsq - (t + ke * LN2_LO)) - f);
}

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

#endif /* UAM_PMATH_H */
