// wm_ndt_ctl.hpp -- the scalar side of pcl::NormalDistributionsTransform::computeTransformation: the Newton step
// (Eigen's JacobiSVD solve of the 6 x 6 system), the More-Thuente line search (computeStepLengthMT,
// trialValueSelectionMT, updateIntervalMT) and the loop around them.  Written once for the host (one registration on
// the whole device: a derivative pass is a kernel, wm_ndt.hip) and the device (one registration per workgroup: wave 0
// runs this, a derivative pass is a workgroup-wide sum, wm_ndt_small.hip).
// Eval: double eval(const double p[6], double *grad, double *hess) -- score (+ gradient, + Hessian; either may be
// null) at pose p; bool failed(); bool skip_line_search(); bool spec_hessian(); void note_line_search(int trials).
// [PCL registration/impl/ndt.hpp; More & Thuente 1994]
#pragma once
#include <float.h>
#include <math.h>

#include "wm_math.hpp"

namespace wm {

// One-sided Jacobi SVD solve x = V S^+ U^T b for a 6x6 system (Eigen JacobiSVD::solve)
WM_HD void svd_solve6(const double *A, const double *b, double *x) {
    constexpr int N = 6;
    double W[N * N], V[N * N];
    for (int i = 0; i < N * N; ++i) W[i] = A[i];
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) V[i * N + j] = (i == j);
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int i = 0; i < N - 1; ++i)
            for (int j = i + 1; j < N; ++j) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int k = 0; k < N; ++k) {
                    alpha += W[k * N + i] * W[k * N + i];
                    beta += W[k * N + j] * W[k * N + j];
                    gamma += W[k * N + i] * W[k * N + j];
                }
                if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 2.3e-16 * sqrt(alpha * beta)) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int k = 0; k < N; ++k) {
                    const double wi = W[k * N + i], wj = W[k * N + j];
                    W[k * N + i] = c * wi - s * wj;
                    W[k * N + j] = s * wi + c * wj;
                    const double vi = V[k * N + i], vj = V[k * N + j];
                    V[k * N + i] = c * vi - s * vj;
                    V[k * N + j] = s * vi + c * vj;
                }
            }
        if (!rotated) break;
    }
    double sv[N], smax = 0;
    for (int j = 0; j < N; ++j) {
        double s2 = 0;
        for (int k = 0; k < N; ++k) s2 += W[k * N + j] * W[k * N + j];
        sv[j] = sqrt(s2);
        if (sv[j] > smax) smax = sv[j];
    }
    const double thr = smax * N * 2.220446049250313e-16;
    double y[N];
    for (int j = 0; j < N; ++j) {
        double s = 0;
        for (int k = 0; k < N; ++k) s += W[k * N + j] * b[k];  // (U S)_j . b
        y[j] = (sv[j] > thr) ? s / (sv[j] * sv[j]) : 0.0;      // U_j . b / S_j
    }
    for (int i = 0; i < N; ++i) {
        double s = 0;
        for (int j = 0; j < N; ++j) s += V[i * N + j] * y[j];
        x[i] = s;
    }
}

WM_HD double psi_mt(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }
WM_HD double dpsi_mt(double g_a, double g_0, double mu) { return g_a - mu * g_0; }

WM_HD bool update_interval_mt(double &a_l, double &f_l, double &g_l, double &a_u, double &f_u,
                               double &g_u, double a_t, double f_t, double g_t) {
    if (f_t > f_l) {
        a_u = a_t; f_u = f_t; g_u = g_t;
        return false;
    } else if (g_t * (a_l - a_t) > 0) {
        a_l = a_t; f_l = f_t; g_l = g_t;
        return false;
    } else if (g_t * (a_l - a_t) < 0) {
        a_u = a_l; f_u = f_l; g_u = g_l;
        a_l = a_t; f_l = f_t; g_l = g_t;
        return false;
    }
    return true;
}

WM_HD double trial_value_mt(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u,
                             double a_t, double f_t, double g_t) {
    if (f_t > f_l) {
        const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
        const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
        return fabs(a_c - a_l) < fabs(a_q - a_l) ? a_c : 0.5 * (a_q + a_c);
    } else if (g_t * g_l < 0) {
        const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
        const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        return fabs(a_c - a_t) >= fabs(a_s - a_t) ? a_c : a_s;
    } else if (fabs(g_t) <= fabs(g_l)) {
        const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
        const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        const double a_n = fabs(a_c - a_t) < fabs(a_s - a_t) ? a_c : a_s;
        return a_t > a_l ? fmin(a_t + 0.66 * (a_u - a_t), a_n) : fmax(a_t + 0.66 * (a_u - a_t), a_n);
    }
    const double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u, w = sqrt(z * z - g_t * g_u);
    return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
}

template <class Eval>
WM_HD double step_length_mt(Eval &E, const double x[6], double dir[6], double step_init,
                            double step_max, double step_min, double *score, double grad[6],
                            double hess[36]) {
    const double phi_0 = -(*score), mu = 1.e-4, nu = 0.9;
    double d_phi_0 = 0, x_t[6];
    for (int a = 0; a < 6; ++a) d_phi_0 -= grad[a] * dir[a];
    if (d_phi_0 >= 0) {
        if (d_phi_0 == 0) return 0;
        d_phi_0 *= -1;
        for (int a = 0; a < 6; ++a) dir[a] *= -1;
    }
    const int max_step_iterations = 10;
    int step_iterations = 0;
    bool hess_at_xt = false;  // `hess` already holds the Hessian at the last trial point
    double a_l = 0, a_u = 0;
    double f_l = psi_mt(a_l, phi_0, phi_0, d_phi_0, mu), g_l = dpsi_mt(d_phi_0, d_phi_0, mu);
    double f_u = psi_mt(a_u, phi_0, phi_0, d_phi_0, mu), g_u = dpsi_mt(d_phi_0, d_phi_0, mu);
    bool interval_converged = E.skip_line_search() ? ((step_max - step_min) > 0) : ((step_max - step_min) < 0);
    bool open_interval = true;
    double a_t = fmax(fmin(step_init, step_max), step_min);
    for (int a = 0; a < 6; ++a) x_t[a] = x[a] + dir[a] * a_t;
    *score = E.eval(x_t, grad, hess);
    if (E.failed()) return 0;
    double phi_t = -(*score), d_phi_t = 0;
    for (int a = 0; a < 6; ++a) d_phi_t -= grad[a] * dir[a];
    double psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu), d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
    while (!interval_converged && step_iterations < max_step_iterations &&
           !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
        a_t = open_interval ? trial_value_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t)
                            : trial_value_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
        a_t = fmax(fmin(a_t, step_max), step_min);
        for (int a = 0; a < 6; ++a) x_t[a] = x[a] + dir[a] * a_t;
        // PCL evaluates score + gradient here and, once the search has ended, the Hessian at the accepted
        // point in a pass of its own (computeHessian).  Nearly every search that gets here ends with THIS
        // trial, and a pass that forms the Hessian costs the same with or without the gradient (150 vs
        // 153 us at 2M points, against 81 us for the gradient alone): so the Hessian is formed along with
        // the FIRST extra trial, and the separate pass (+ its round trip) is dropped when that trial is
        // accepted.  Same sums over the same terms as computeHessian's; a rejected trial wastes 72 us, and
        // a search that rejects its first extra trial usually goes on for many (on the bench pair: 13
        // searches without an extra trial, 9 with one, 1 with ten), so later trials are not speculated on.
        const bool spec = E.spec_hessian() && step_iterations == 0;
        *score = E.eval(x_t, grad, spec ? hess : nullptr);
        hess_at_xt = spec;
        if (E.failed()) return 0;
        phi_t = -(*score);
        d_phi_t = 0;
        for (int a = 0; a < 6; ++a) d_phi_t -= grad[a] * dir[a];
        psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu);
        d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
        if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
            open_interval = false;
            f_l = f_l + phi_0 - mu * d_phi_0 * a_l;
            g_l = g_l + mu * d_phi_0;
            f_u = f_u + phi_0 - mu * d_phi_0 * a_u;
            g_u = g_u + mu * d_phi_0;
        }
        interval_converged = open_interval
                                 ? update_interval_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t)
                                 : update_interval_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
        ++step_iterations;
    }
    E.note_line_search(step_iterations);
    if (step_iterations && !hess_at_xt) {  // computeHessian at the accepted point
        (void) E.eval(x_t, nullptr, hess);
    }
    return a_t;
}

struct NdtLoopOut {
    double p[6];
    double score;
    int iterations;
    bool converged;
};

// computeTransformation from the identity guess: p = (x, y, z, roll, pitch, yaw)
template <class Eval>
WM_HD void ndt_align_loop(Eval &E, double step_size, double t_eps, int max_iter, int force_iterations, NdtLoopOut *out) {
    double p[6] = {0, 0, 0, 0, 0, 0}, grad[6], hess[36], delta[6];
    int iter = 0;
    bool converged = false;
    const int max_it = force_iterations > 0 ? force_iterations : max_iter;
    double score = E.eval(p, grad, hess);
    while (!converged && !E.failed()) {
        double neg[6], norm = 0;
        for (int a = 0; a < 6; ++a) neg[a] = -grad[a];
        svd_solve6(hess, neg, delta);
        for (int a = 0; a < 6; ++a) norm += delta[a] * delta[a];
        norm = sqrt(norm);
        if (norm == 0 || norm != norm) {
            converged = (norm == norm);
            break;
        }
        for (int a = 0; a < 6; ++a) delta[a] /= norm;
        const double alpha = step_length_mt(E, p, delta, norm, step_size, t_eps / 2, &score, grad, hess);
        if (E.failed()) break;
        for (int a = 0; a < 6; ++a) p[a] += delta[a] * alpha;
        if (force_iterations > 0) {
            if (iter + 1 >= max_it) converged = true;
        } else if (iter > max_it || (iter && fabs(alpha) < t_eps)) {
            converged = true;
        }
        ++iter;
    }
    for (int a = 0; a < 6; ++a) out->p[a] = p[a];
    out->score = score;
    out->iterations = iter;
    out->converged = converged;
}

}  // namespace wm
