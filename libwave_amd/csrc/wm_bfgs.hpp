// wm_bfgs.hpp -- pcl::BFGS as GICP's estimateRigidTransformationBFGS drives it (GSL's vector_bfgs2 with Fletcher's
// line search), and applyState / the rotation part of the gradient around it.  Scalar code on six unknowns, written
// once for the host (one registration on the whole device: the objective is evaluated by a kernel, wm_gicp.hip) and
// the device (one registration per workgroup: every thread runs it, the objective is a workgroup-wide sum,
// wm_gicp_small.hip).  Fn: double fdf(const double x[6], double g[6]); int pairs(); bool failed().
// [PCL registration/bfgs.h, registration/impl/gicp.hpp]
//
// THIRD-PARTY PROVENANCE (none of it is in /root/reference; all restated from published algorithms, no source copied):
//   * bfgs_minimize / the line search: PCL 1.8's BFGS (BSD-3), itself a C++ rendering of GSL's vector_bfgs2 and
//     Fletcher's line search (GNU Scientific Library, multimin/vector_bfgs2.c, linear_minimize.c);
//   * libm_sincosf: the sinf / cosf of glibc >= 2.28, i.e. ARM's "optimized-routines" single-precision algorithm
//     (MIT / LGPL; constants = its published minimax coefficients);
//   * libm_atanf / libm_atan2f / libm_asinf: fdlibm's float functions (Sun Microsystems, "Freely distributable"
//     notice; s_atanf.c, e_atan2f.c, e_asinf.c as glibc 2.35 ships them).
// Why they are here at all: so that the batched path and the one-pair path (host libm) take the same branches
// (section 4.5 of DESIGN.md) -- a self-consistency measure of this repo, not a parity requirement of the reference.
#pragma once
#include <float.h>
#include <math.h>

#include "wm_math.hpp"

namespace wm {

// sinf / cosf with glibc's results (2.28 and later: the algorithm of ARM's optimized routines -- the argument in
// double, reduced by multiples of pi/2 with one multiplication, a degree-7 / degree-8 polynomial in double, ONE
// rounding to float).  PCL's applyState builds the float transform from cosf / sinf of the three angles, the host
// path calls libm for them, and the objective sees every bit of that transform: a device-library sinf that is one
// ulp off in 1 % of its calls sends the optimiser down another branch within a few evaluations (measured: the two
// paths then stop up to 1.5 mm apart -- both within BFGS's tolerance, but not the same answer).  Restated here from
// the published algorithm and its constants; tests/test_bfgs_trig_cpu.py compares it with the installed libm on 40
// million arguments (2-3 differ, where libm's FMA build rounds an intermediate differently).  |x| < 120 only:
// beyond, the device library's function (GICP's angles are Euler angles of a registration step).
struct SincosfTab {
    double c0, c1, c2, c3, c4, s1, s2, s3;
};
WM_HD float libm_sincosf_poly(double x, double x2, double sgn, int n) {
    // (the table for an odd multiple of pi: the cosine's coefficients negated)
    const SincosfTab p = {sgn * 0x1p0, sgn * -0x1.ffffffd0c621cp-2, sgn * 0x1.55553e1068f19p-5, sgn * -0x1.6c087e89a359dp-10,
                          sgn * 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13};
    if ((n & 1) == 0) {
        const double x3 = x * x2, s1 = p.s2 + x2 * p.s3, x7 = x3 * x2, s = x + x3 * p.s1;
        return (float) (s + x7 * s1);
    }
    const double x4 = x2 * x2, c2 = p.c3 + x2 * p.c4, c1 = p.c0 + x2 * p.c1, x6 = x4 * x2, c = c1 + x4 * p.c2;
    return (float) (c + x6 * c2);
}
WM_HD float libm_sincosf(float y, int cosine) {
    const float ay = fabsf(y);
    double x = (double) y;
    if (ay < 0x1.8p-1f) {  // (the library compares the top 12 bits of |y| with those of pi/4: that is |y| < 0.75)
        if (ay < 0x1p-12f) return cosine ? 1.0f : y;
        return libm_sincosf_poly(x, x * x, 1.0, cosine);
    }
    if (!(ay < 120.0f)) return cosine ? cosf(y) : sinf(y);
    const double r = x * 0x1.45F306DC9C883p+23;
    const int n = ((int) r + 0x800000) >> 24;
    x = x - (double) n * 0x1.921FB54442D18p0;
    const double sign = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return libm_sincosf_poly(x * sign, x * x, (n & 2) ? -1.0 : 1.0, n ^ cosine);
}

// atan2f / asinf with glibc's results: PCL reads the three Euler angles back from the float transform with
// atan2(m21, m22), asin(-m20), atan2(m10, m00) -- float arguments, so the FLOAT functions -- at the start of every
// outer iteration.  glibc's (2.35) are the fdlibm float versions -- a quarter of atan2f's results are not the
// correctly rounded ones, so "compute in double and round" does not reproduce them.  Restated here from the
// published algorithm (argument reduction to four intervals, the degree-10 odd polynomial, hi/lo constants), in plain
// float arithmetic without contraction, as the generic build of libm evaluates it.
// tests/test_bfgs_trig_cpu.py: no difference from the installed libm on six million arguments each, all ranges.
WM_HD unsigned libm_f2u(float f) {
    unsigned u;
    __builtin_memcpy(&u, &f, 4);
    return u;
}
WM_HD float libm_u2f(unsigned u) {
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
WM_HD float libm_atanf(float x) {
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f,  -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                          6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    const unsigned hx = libm_f2u(x), ix = hx & 0x7fffffffu;
    int id;
    if (ix >= 0x4c000000u) {  // |x| >= 2^25
        if (ix > 0x7f800000u) return x + x;
        return (hx >> 31) ? -atanhi[3] - atanlo[3] : atanhi[3] + atanlo[3];
    }
    if (ix < 0x3ee00000u) {  // |x| < 0.4375
        if (ix < 0x31000000u) return x;
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000u) {      // |x| < 1.1875
            if (ix < 0x3f300000u) {  // 7/16 <= |x| < 11/16
                id = 0;
                x = (2.0f * x - 1.0f) / (2.0f + x);
            } else {
                id = 1;
                x = (x - 1.0f) / (x + 1.0f);
            }
        } else if (ix < 0x401c0000u) {  // |x| < 2.4375
            id = 2;
            x = (x - 1.5f) / (1.0f + 1.5f * x);
        } else {
            id = 3;
            x = -1.0f / x;
        }
    }
    const float z = x * x, w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return (hx >> 31) ? -r : r;
}
WM_HD float libm_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int hx = (int) libm_f2u(x), hy = (int) libm_f2u(y);
    const int ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return libm_atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000 || iy == 0x7f800000) return atan2f(y, x);  // (an infinite entry: no transform has one)
    float z;
    const int k = (iy - ix) >> 23;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = libm_atanf(fabsf(y / x));
    switch (m) {
        case 0: return z;
        case 1: return -z;
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}
WM_HD float libm_asinf(float x) {
    const float pio2_hi = 1.57079637050628662109375f, pio2_lo = -4.37113900018624283e-8f, pio4_hi = 0.785398185253143310546875f;
    const float p0 = 1.666675248e-1f, p1 = 7.495297643e-2f, p2 = 4.547037598e-2f, p3 = 2.417951451e-2f, p4 = 4.216630880e-2f;
    const int hx = (int) libm_f2u(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return x * pio2_hi + x * pio2_lo;  // |x| = 1
    if (ix > 0x3f800000) return (x - x) / (x - x);           // |x| > 1: NaN
    if (ix < 0x3f000000) {                                   // |x| < 0.5
        if (ix < 0x32000000) return x;
        const float t = x * x, w = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
        return x + x * w;
    }
    float w = 1.0f - fabsf(x), t = w * 0.5f;
    float p = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
    const float s = sqrtf(t);
    if (ix >= 0x3F79999A) {  // |x| > 0.975
        t = pio2_hi - (2.0f * (s + s * p) - pio2_lo);
    } else {
        w = libm_u2f(libm_f2u(s) & 0xfffff000u);
        const float c = (t - w * w) / (s + w), r = p;
        p = 2.0f * s * r - (pio2_lo - 2.0f * c);
        const float q = pio4_hi - 2.0f * w;
        t = pio4_hi - (p - q);
    }
    return hx > 0 ? t : -t;
}

WM_HD void state_to_matrix_f(const double base[16], const double x[6], float T[16]) {
#ifdef __HIP_DEVICE_COMPILE__
    const float cphi = libm_sincosf((float) x[3], 1), sphi = libm_sincosf((float) x[3], 0);
    const float cth = libm_sincosf((float) x[4], 1), sth = libm_sincosf((float) x[4], 0);
    const float cpsi = libm_sincosf((float) x[5], 1), spsi = libm_sincosf((float) x[5], 0);
#else
    const float cphi = cosf((float) x[3]), sphi = sinf((float) x[3]);
    const float cth = cosf((float) x[4]), sth = sinf((float) x[4]);
    const float cpsi = cosf((float) x[5]), spsi = sinf((float) x[5]);
#endif
    float R[9], B[16], o[16];
    R[0] = cpsi * cth;
    R[1] = cpsi * sth * sphi - spsi * cphi;
    R[2] = cpsi * sth * cphi + spsi * sphi;
    R[3] = spsi * cth;
    R[4] = spsi * sth * sphi + cpsi * cphi;
    R[5] = spsi * sth * cphi - cpsi * sphi;
    R[6] = -sth;
    R[7] = cth * sphi;
    R[8] = cth * cphi;
    for (int i = 0; i < 16; ++i) B[i] = (float) base[i];
    for (int i = 0; i < 16; ++i) o[i] = B[i];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = 0;
            for (int k = 0; k < 3; ++k) s += R[i * 3 + k] * B[k * 4 + j];
            o[i * 4 + j] = s;
        }
    o[3] = B[3] + (float) x[0];
    o[7] = B[7] + (float) x[1];
    o[11] = B[11] + (float) x[2];
    for (int i = 0; i < 16; ++i) T[i] = o[i];
}

WM_HD void r_derivative(const double x[6], const double Racc[9], double g[6]) {
    const double phi = x[3], theta = x[4], psi = x[5];
    const double cphi = cos(phi), sphi = sin(phi), ctheta = cos(theta), stheta = sin(theta),
                 cpsi = cos(psi), spsi = sin(psi);
    const double dPhi[9] = {0, sphi * spsi + cphi * cpsi * stheta, cphi * spsi - cpsi * sphi * stheta,
                            0, -cpsi * sphi + cphi * spsi * stheta, -cphi * cpsi - sphi * spsi * stheta,
                            0, cphi * ctheta, -ctheta * sphi};
    const double dTheta[9] = {-cpsi * stheta, cpsi * ctheta * sphi, cphi * cpsi * ctheta,
                              -spsi * stheta, ctheta * sphi * spsi, cphi * ctheta * spsi,
                              -ctheta, -sphi * stheta, -cphi * stheta};
    const double dPsi[9] = {-ctheta * spsi, -cphi * cpsi - sphi * spsi * stheta, cpsi * sphi - cphi * spsi * stheta,
                            cpsi * ctheta, -cphi * spsi + cpsi * sphi * stheta, sphi * spsi + cphi * cpsi * stheta,
                            0, 0, 0};
    g[3] = g[4] = g[5] = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {  // matricesInnerProd: sum mat1(j,i) * mat2(i,j)
            g[3] += dPhi[j * 3 + i] * Racc[i * 3 + j];
            g[4] += dTheta[j * 3 + i] * Racc[i * 3 + j];
            g[5] += dPsi[j * 3 + i] * Racc[i * 3 + j];
        }
}


// ---- pcl::BFGS (GSL vector_bfgs2 + Fletcher line search)
template <class Fn>
struct LineFn {
    Fn *F;
    double x0[6], p[6], f0, df0;
    double x_a[6], g_a[6], alpha_c, f_c, df_c;
    bool have_c = false;
    WM_HD void eval(double alpha) {
        if (have_c && alpha == alpha_c) return;
        for (int i = 0; i < 6; ++i) x_a[i] = x0[i] + alpha * p[i];
        f_c = F->fdf(x_a, g_a);
        df_c = 0;
        for (int i = 0; i < 6; ++i) df_c += g_a[i] * p[i];
        alpha_c = alpha;
        have_c = true;
    }
};

WM_HD double cubic(double c0, double c1, double c2, double c3, double z) { return c0 + z * (c1 + z * (c2 + z * c3)); }
WM_HD void check_extremum(double c0, double c1, double c2, double c3, double z, double *zmin, double *fmin) {
    const double y = cubic(c0, c1, c2, c3, z);
    if (y < *fmin) {
        *zmin = z;
        *fmin = y;
    }
}
WM_HD int solve_quadratic(double a, double b, double c, double *x0, double *x1) {
    if (a == 0) {
        if (b == 0) return 0;
        *x0 = -c / b;
        return 1;
    }
    const double disc = b * b - 4 * a * c;
    if (disc > 0) {
        if (b == 0) {
            const double r = sqrt(-c / a);
            *x0 = -r;
            *x1 = r;
        } else {
            const double sgnb = (b > 0 ? 1 : -1);
            const double temp = -0.5 * (b + sgnb * sqrt(disc));
            const double r1 = temp / a, r2 = c / temp;
            *x0 = r1 < r2 ? r1 : r2;
            *x1 = r1 < r2 ? r2 : r1;
        }
        return 2;
    } else if (disc == 0) {
        *x0 = *x1 = -0.5 * b / a;
        return 2;
    }
    return 0;
}
WM_HD double interp_quad(double f0, double fp0, double f1, double zl, double zh) {
    const double fl = f0 + zl * (fp0 + zl * (f1 - f0 - fp0));
    const double fh = f0 + zh * (fp0 + zh * (f1 - f0 - fp0));
    const double c = 2 * (f1 - f0 - fp0);
    double zmin = zl, fmin = fl;
    if (fh < fmin) {
        zmin = zh;
        fmin = fh;
    }
    if (c > 0) {
        const double z = -fp0 / c;
        if (z > zl && z < zh) {
            const double f = f0 + z * (fp0 + z * (f1 - f0 - fp0));
            if (f < fmin) {
                zmin = z;
                fmin = f;
            }
        }
    }
    return zmin;
}
WM_HD double interp_cubic(double f0, double fp0, double f1, double fp1, double zl, double zh) {
    const double eta = 3 * (f1 - f0) - 2 * fp0 - fp1, xi = fp0 + fp1 - 2 * (f1 - f0);
    const double c0 = f0, c1 = fp0, c2 = eta, c3 = xi;
    double zmin = zl, fmin = cubic(c0, c1, c2, c3, zl), z0, z1;
    check_extremum(c0, c1, c2, c3, zh, &zmin, &fmin);
    const int n = solve_quadratic(3 * c3, 2 * c2, c1, &z0, &z1);
    if (n == 2) {
        if (z0 > zl && z0 < zh) check_extremum(c0, c1, c2, c3, z0, &zmin, &fmin);
        if (z1 > zl && z1 < zh) check_extremum(c0, c1, c2, c3, z1, &zmin, &fmin);
    } else if (n == 1) {
        if (z0 > zl && z0 < zh) check_extremum(c0, c1, c2, c3, z0, &zmin, &fmin);
    }
    return zmin;
}
WM_HD double interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin,
                          double xmax, int order) {
    double ymin = (xmin - a) / (b - a), ymax = (xmax - a) / (b - a);
    if (ymin > ymax) {
        const double t = ymin;
        ymin = ymax;
        ymax = t;
    }
    const double y = (order > 2 && fpb == fpb) ? interp_cubic(fa, fpa * (b - a), fb, fpb * (b - a), ymin, ymax)
                                               : interp_quad(fa, fpa * (b - a), fb, ymin, ymax);
    return a + y * (b - a);
}

template <class Fn>
WM_HD bool line_search(LineFn<Fn> &L, double rho, double sigma, double tau1, double tau2, double tau3,
                        int order, double alpha1, double *alpha_new) {
    const double f0 = L.f0, fp0 = L.df0;
    double falpha, falpha_prev = f0, fpalpha, fpalpha_prev = fp0, delta, alpha_next;
    double alpha = alpha1, alpha_prev = 0.0;
    double a = 0.0, b = alpha, fa = f0, fb = 0.0, fpa = fp0, fpb = 0.0;
    int i = 0;
    while (i++ < 100) {
        L.eval(alpha);
        falpha = L.f_c;
        if (falpha > f0 + alpha * rho * fp0 || falpha >= falpha_prev) {
            a = alpha_prev; fa = falpha_prev; fpa = fpalpha_prev;
            b = alpha; fb = falpha; fpb = NAN;
            break;
        }
        fpalpha = L.df_c;
        if (fabs(fpalpha) <= -sigma * fp0) {
            *alpha_new = alpha;
            return true;
        }
        if (fpalpha >= 0) {
            a = alpha; fa = falpha; fpa = fpalpha;
            b = alpha_prev; fb = falpha_prev; fpb = fpalpha_prev;
            break;
        }
        delta = alpha - alpha_prev;
        alpha_next = interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha,
                                 alpha + delta, alpha + tau1 * delta, order);
        alpha_prev = alpha; falpha_prev = falpha; fpalpha_prev = fpalpha;
        alpha = alpha_next;
    }
    while (i++ < 100) {
        delta = b - a;
        alpha = interpolate(a, fa, fpa, b, fb, fpb, a + tau2 * delta, b - tau3 * delta, order);
        L.eval(alpha);
        falpha = L.f_c;
        if ((a - alpha) * fpa <= DBL_EPSILON) return false;  // roundoff prevents progress
        if (falpha > f0 + rho * alpha * fp0 || falpha >= fa) {
            b = alpha; fb = falpha; fpb = NAN;
        } else {
            fpalpha = L.df_c;
            if (fabs(fpalpha) <= -sigma * fp0) {
                *alpha_new = alpha;
                return true;
            }
            if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) {
                b = a; fb = fa; fpb = fpa;
                a = alpha; fa = falpha; fpa = fpalpha;
            } else {
                a = alpha; fa = falpha; fpa = fpalpha;
            }
        }
    }
    *alpha_new = alpha;
    return true;
}

WM_HD double norm6(const double *v) {
    double s = 0;
    for (int i = 0; i < 6; ++i) s += v[i] * v[i];
    return sqrt(s);
}

// estimateRigidTransformationBFGS; returns inner iterations, -1 if < 4 pairs
template <class Fn>
WM_HD int bfgs_minimize(Fn &F, double x[6], int max_inner, double *f_out) {
    const double rho = 0.01, sigma = 0.01, tau1 = 9, tau2 = 0.05, tau3 = 0.5, gradient_tol = 1e-2;
    if (F.pairs() < 4) return -1;
    double g[6], x0[6], g0[6], p[6], dx0[6], dg0[6];
    double f = F.fdf(x, g);
    for (int i_ = 0; i_ < 6; ++i_) x0[i_] = x[i_];
    for (int i_ = 0; i_ < 6; ++i_) g0[i_] = g[i_];
    double g0norm = norm6(g0);
    // F.test_at_start() (the statistics objective only, wm_gicp_quad.hpp): the minimiser's own success test is made at
    // the starting point too.  pcl::BFGS tests only AFTER a step; with PCL's per-pair objective a step from an already
    // converged point dies in the float dust of the objective (NoProgress: x unchanged, the outer loop stops).  The
    // statistics have no dust: the step succeeds, moves x by ~1e-6, the float transform changes in its last bit, and
    // the outer loop (libwave's r_eps = 1e-8) re-pairs and crawls on for dozens of iterations that change nothing.
    if (F.test_at_start() && g0norm < 1e-2) {
        if (f_out) *f_out = f;
        return 0;
    }
    for (int i = 0; i < 6; ++i) p[i] = -g0[i] / g0norm;
    double pnorm = norm6(p), fp0 = -g0norm, delta_f = 0;
    int inner = 0;
    do {
        ++inner;
        if (F.failed()) break;
        if (pnorm == 0.0 || g0norm == 0.0 || fp0 == 0 || pnorm != pnorm || g0norm != g0norm) break;
        const double f_prev = f;
        double alpha = 0, alpha1;
        if (delta_f < 0) {
            const double del = fmax(-delta_f, 10 * DBL_EPSILON * fabs(f_prev));
            alpha1 = fmin(1.0, 2.0 * del / (-fp0));
        } else {
            alpha1 = 1.0;  // parameters.step_size
        }
        LineFn<Fn> L;
        L.F = &F;
        for (int i_ = 0; i_ < 6; ++i_) L.x0[i_] = x0[i_];
        for (int i_ = 0; i_ < 6; ++i_) L.p[i_] = p[i_];
        L.f0 = f_prev;
        L.df0 = fp0;
        if (!line_search(L, rho, sigma, tau1, tau2, tau3, 3, alpha1, &alpha)) break;
        L.eval(alpha);
        for (int i_ = 0; i_ < 6; ++i_) x[i_] = L.x_a[i_];
        for (int i_ = 0; i_ < 6; ++i_) g[i_] = L.g_a[i_];
        f = L.f_c;
        delta_f = f - f_prev;
        double dxg = 0, dgg = 0, dxdg = 0, A, B, pg = 0;
        for (int i = 0; i < 6; ++i) {
            dx0[i] = x[i] - x0[i];
            dg0[i] = g[i] - g0[i];
        }
        for (int i = 0; i < 6; ++i) {
            dxg += dx0[i] * g[i];
            dgg += dg0[i] * g[i];
            dxdg += dx0[i] * dg0[i];
        }
        const double dgnorm = norm6(dg0);
        if (dxdg != 0) {
            B = dxg / dxdg;
            A = -(1.0 + dgnorm * dgnorm / dxdg) * B + dgg / dxdg;
        } else {
            B = 0;
            A = 0;
        }
        for (int i = 0; i < 6; ++i) p[i] = g[i] - A * dx0[i] - B * dg0[i];
        for (int i_ = 0; i_ < 6; ++i_) g0[i_] = g[i_];
        for (int i_ = 0; i_ < 6; ++i_) x0[i_] = x[i_];
        g0norm = norm6(g0);
        pnorm = norm6(p);
        for (int i = 0; i < 6; ++i) pg += p[i] * g0[i];
        const double dir = (pg >= 0) ? -1.0 : +1.0;
        for (int i = 0; i < 6; ++i) p[i] *= dir / pnorm;
        pnorm = norm6(p);
        fp0 = 0;
        for (int i = 0; i < 6; ++i) fp0 += p[i] * g0[i];
        if (norm6(g) < gradient_tol) break;
    } while (inner < max_inner);
    if (f_out) *f_out = f;
    return inner;
}


}  // namespace wm
