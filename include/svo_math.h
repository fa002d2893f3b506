/* svo_math.h -- the handful of transcendental functions the geometry solvers call, as ONE software
 * implementation shared by the HIP library and the CPU oracle.
 *
 * Why: sin / cos / acos / cbrt / log decide thresholds on the pose path (the cubic of the 7-point
 * solver, Rodrigues and its inverse in the PnP refinement, the adaptive RANSAC bound).  The device
 * math library and the host's libm round these differently in the last place, which is enough to flip
 * an inlier at a threshold some tens of frames into a sequence.  With the functions below both sides
 * execute the same IEEE-754 double operations (+, -, *, /, sqrt: correctly rounded on the host and on
 * gfx950; both builds use -ffp-contract=off, so no multiply-add is ever fused) in the same order, and
 * agree bit for bit.
 *
 * What: argument reduction + polynomial kernels of the classic kind (Cody-Waite reduction by pi/2 in
 * three pieces, minimax polynomials for sin / cos on [-pi/4, pi/4], a rational approximation for
 * asin on [0, 1/2], s = f / (2 + f) series for log).  Accuracy against the correctly rounded result is
 * about 1 ulp (tests/test_svo_math.py measures it against numpy); it is NOT a claim to equal any
 * particular libm -- upstream OpenCV / g2o link whatever libm their platform has.
 *
 * Plain C99, also valid C++ / HIP; SVO_MATH_FN marks the functions __host__ __device__ under hipcc.
 */
#ifndef SVO_MATH_H
#define SVO_MATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define SVO_MATH_FN static __host__ __device__ __inline__
#define SVO_MATH_SQRT(x) __builtin_sqrt(x)
#define SVO_MATH_RINT(x) __builtin_rint(x)
#define SVO_MATH_FABS(x) __builtin_fabs(x)
#else
#include <math.h>
#define SVO_MATH_FN static inline
#define SVO_MATH_SQRT(x) sqrt(x)
#define SVO_MATH_RINT(x) rint(x)
#define SVO_MATH_FABS(x) fabs(x)
#endif

SVO_MATH_FN uint64_t svo_m_bits(double x)
{
    uint64_t u;
    memcpy(&u, &x, sizeof(u));
    return u;
}
SVO_MATH_FN double svo_m_from_bits(uint64_t u)
{
    double x;
    memcpy(&x, &u, sizeof(x));
    return x;
}

/* sin and cos on [-pi/4, pi/4]: odd / even minimax polynomials in z = x^2, Horner, no fused operations */
SVO_MATH_FN double svo_m_ksin(double x)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x + (z * x) * (S1 + z * r);
}
SVO_MATH_FN double svo_m_kcos(double x)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + z * r);
}

/* x = k * pi/2 + r, |r| <= pi/4 (+ a rounding): returns r, *quadrant = k mod 4.  |x| < 2^20 * pi/2 keeps
 * every product with the 33-bit pieces of pi/2 exact; larger arguments (never met on this path) lose
 * accuracy gracefully, they do not lose determinism. */
SVO_MATH_FN double svo_m_reduce(double x, int *quadrant)
{
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double P1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
    const double P2 = 6.07710050630396597660e-11;  /* next 33 bits */
    const double P3 = 2.02226624871116645580e-21;  /* next 33 bits */
    const double P3T = 8.47842766036889956997e-32; /* pi/2 - (P1 + P2 + P3) */
    const double k = SVO_MATH_RINT(x * INV_PIO2);
    double r = x - k * P1;
    r = r - k * P2;
    r = r - k * P3;
    r = r - k * P3T;
    *quadrant = (int)((long long)k & 3);
    return r;
}

SVO_MATH_FN double svo_sin(double x)
{
    int q;
    const double r = svo_m_reduce(x, &q);
    switch (q) {
    case 0:
        return svo_m_ksin(r);
    case 1:
        return svo_m_kcos(r);
    case 2:
        return -svo_m_ksin(r);
    default:
        return -svo_m_kcos(r);
    }
}

SVO_MATH_FN double svo_cos(double x)
{
    int q;
    const double r = svo_m_reduce(x, &q);
    switch (q) {
    case 0:
        return svo_m_kcos(r);
    case 1:
        return -svo_m_ksin(r);
    case 2:
        return -svo_m_kcos(r);
    default:
        return svo_m_ksin(r);
    }
}

/* (asin(x) - x) / x^3 for z = x^2 in [0, 1/4]: rational approximation p(z) / q(z) */
SVO_MATH_FN double svo_m_asin_r(double z)
{
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 = 7.70381505559019352791e-02;
    const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    return p / q;
}

/* acos on [-1, 1]; arguments outside are clamped (callers clamp already: a cosine from a rotation trace) */
SVO_MATH_FN double svo_acos(double x)
{
    const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17;
    const double PI = 3.14159265358979311600e+00;
    if (x >= 1.0)
        return 0.0;
    if (x <= -1.0)
        return PI + 2.0 * PIO2_LO;
    const double ax = SVO_MATH_FABS(x);
    if (ax < 0.5) {
        if (ax < 5.55111512312578270212e-17) /* 2^-54: acos(x) rounds to pi/2 */
            return PIO2_HI + PIO2_LO;
        const double r = svo_m_asin_r(x * x);
        return PIO2_HI - (x - (PIO2_LO - x * r));
    }
    if (x < 0) { /* acos(x) = pi - 2 asin(sqrt((1 + x) / 2)) */
        const double z = (1.0 + x) * 0.5;
        const double s = SVO_MATH_SQRT(z);
        const double r = svo_m_asin_r(z);
        const double w = r * s - PIO2_LO;
        return PI - 2.0 * (s + w);
    }
    { /* acos(x) = 2 asin(sqrt((1 - x) / 2)), the square root split into a 21-bit head and a correction */
        const double z = (1.0 - x) * 0.5;
        const double s = SVO_MATH_SQRT(z);
        const double df = svo_m_from_bits(svo_m_bits(s) & 0xffffffff00000000ull);
        const double c = (z - df * df) / (s + df);
        const double r = svo_m_asin_r(z);
        const double w = r * s + c;
        return 2.0 * (df + w);
    }
}

/* cube root: exponent / 3 by integer arithmetic on the bit pattern, then Halley steps (cubic convergence)
 * and one Newton step on the residual; x = 0 and negative arguments handled, no subnormal inputs on this path */
SVO_MATH_FN double svo_cbrt(double x)
{
    if (x == 0.0)
        return x;
    const double ax = SVO_MATH_FABS(x);
    /* high word / 3 + bias: within a few percent of the root */
    const uint64_t hi = (svo_m_bits(ax) >> 32) / 3u + 715094163u;
    double t = svo_m_from_bits(hi << 32);
    for (int i = 0; i < 4; i++) {
        const double t3 = t * t * t;
        t = t * ((t3 + 2.0 * ax) / (2.0 * t3 + ax));
    }
    t = t - (t - ax / (t * t)) * (1.0 / 3.0);
    return x < 0 ? -t : t;
}

/* natural logarithm of a positive, normal double */
SVO_MATH_FN double svo_log(double x)
{
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t u = svo_m_bits(x);
    int k = (int)((u >> 52) & 0x7ff) - 1023;
    u = (u & 0x000fffffffffffffull) | 0x3ff0000000000000ull; /* mantissa in [1, 2) */
    double m = svo_m_from_bits(u);
    if (m > 1.41421356237309514547e+00) { /* keep 1 + f in (sqrt(2) / 2, sqrt(2)] */
        m = m * 0.5;
        k += 1;
    }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * LN2_HI - ((hfsq - (s * (hfsq + R) + dk * LN2_LO)) - f);
}

/* x^n for a small positive integer n, multiplications in ascending order (x * x * ... * x) */
SVO_MATH_FN double svo_powi(double x, int n)
{
    double r = x;
    for (int i = 1; i < n; i++)
        r = r * x;
    return r;
}

#endif /* SVO_MATH_H */
