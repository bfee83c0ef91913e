/*
 * uam_pmath_impl.h — bodies of the portable fp64 sincos / log (see uam_pmath.h).  Included once per
 * flavour with UAM_PM_NAME (function-name decoration) and UAM_KC (how a polynomial coefficient is
 * materialised) defined by the includer, and UAM_FMAK(a, z, c) = fma(a, z, coefficient c) (a Horner step).  No include
 * guard on purpose.
 */
/* sin and cos of x (|x| < ~1e6 for full accuracy).
 * Reduction: k = rint(x*2/pi), y = x - k*(C1+C2+C3) carried as y0+y1 with
 * fma-exact products; kernels are Taylor polynomials on [-pi/4, pi/4]
 * evaluated in the fdlibm "x + x^3*P" / "1 - z/2 + z^2*Q" forms. */
UAM_HD void UAM_PM_NAME(uam_sincos)(double x, double* sn, double* cs)
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
    ps = UAM_FMAK(ps, z, -0x1.2f49b46814157p-57);
    ps = UAM_FMAK(ps, z, 0x1.952c77030ad4ap-49);
    ps = UAM_FMAK(ps, z, -0x1.ae7f3e733b81fp-41);
    ps = UAM_FMAK(ps, z, 0x1.6124613a86d09p-33);
    ps = UAM_FMAK(ps, z, -0x1.ae64567f544e4p-26);
    ps = UAM_FMAK(ps, z, 0x1.71de3a556c734p-19);
    ps = UAM_FMAK(ps, z, -0x1.a01a01a01a01ap-13);
    ps = UAM_FMAK(ps, z, 0x1.1111111111111p-7);   /* ps = S5 + S7 z + ... */
    double v = z * y0;
    /* sin(y0+y1) = y0 + (v*(S3 + z*ps) + y1*(1 - z/2)) */
    double S = y0 + (uam_fma(v, uam_fma(z, ps, UAM_KC(-0x1.5555555555555p-3)), 0.0) + y1 * (1.0 - 0.5 * z));
    /* cos kernel */
    double pc = UAM_KC(-0x1.0ce396db7f853p-70);
    pc = UAM_FMAK(pc, z, 0x1.e542ba4020225p-62);
    pc = UAM_FMAK(pc, z, -0x1.6827863b97d97p-53);
    pc = UAM_FMAK(pc, z, 0x1.ae7f3e733b81fp-45);
    pc = UAM_FMAK(pc, z, -0x1.93974a8c07c9dp-37);
    pc = UAM_FMAK(pc, z, 0x1.1eed8eff8d898p-29);
    pc = UAM_FMAK(pc, z, -0x1.27e4fb7789f5cp-22);
    pc = UAM_FMAK(pc, z, 0x1.a01a01a01a01ap-16);
    pc = UAM_FMAK(pc, z, -0x1.6c16c16c16c17p-10);
    pc = UAM_FMAK(pc, z, 0x1.5555555555555p-5);   /* pc = C4 + C6 z + ... */
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
UAM_HD double UAM_PM_NAME(uam_log)(double x)
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
    r = UAM_FMAK(r, z, 0x1.642c8590b2164p-4);
    r = UAM_FMAK(r, z, 0x1.8618618618618p-4);
    r = UAM_FMAK(r, z, 0x1.af286bca1af28p-4);
    r = UAM_FMAK(r, z, 0x1.e1e1e1e1e1e1ep-4);
    r = UAM_FMAK(r, z, 0x1.1111111111111p-3);
    r = UAM_FMAK(r, z, 0x1.3b13b13b13b14p-3);
    r = UAM_FMAK(r, z, 0x1.745d1745d1746p-3);
    r = UAM_FMAK(r, z, 0x1.c71c71c71c71cp-3);
    r = UAM_FMAK(r, z, 0x1.2492492492492p-2);
    r = UAM_FMAK(r, z, 0x1.999999999999ap-2);
    r = UAM_FMAK(r, z, 0x1.5555555555555p-1);
    /* f - s*(f - z*r): fdlibm form, exact-ish head f */
    double hfsq = 0.5 * f * f;
    double ke = (double)e;
    double t = s * (hfsq + z * r);
    return ke * LN2_HI - ((hfsq - (t + ke * LN2_LO)) - f);
}

