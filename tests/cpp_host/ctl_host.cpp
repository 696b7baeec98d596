// The scalar control code that is compiled for both sides (libwave_amd/csrc/wm_bfgs.hpp, wm_ndt_ctl.hpp), here on the host
// and away from any device: pcl::BFGS on a 6-D quadratic + quartic bowl whose minimum is known, and Eigen's JacobiSVD solve
// (svd_solve6) against a system built from its solution.
#include <cmath>
#include <cstdio>

#include "wm_bfgs.hpp"
#include "wm_ndt_ctl.hpp"

struct Bowl {
    double c[6] = {0.3, -0.2, 0.05, 0.01, -0.02, 0.03};
    double w[6] = {1.0, 2.0, 0.5, 40.0, 25.0, 60.0};
    int evals = 0;
    int pairs() const { return 1000; }
    bool failed() const { return false; }
    bool test_at_start() const { return false; }
    double fdf(const double x[6], double g[6]) {
        ++evals;
        double f = 0;
        for (int i = 0; i < 6; ++i) {
            const double d = x[i] - c[i];
            f += w[i] * (d * d + 0.1 * d * d * d * d);
            if (g) g[i] = w[i] * (2 * d + 0.4 * d * d * d);
        }
        return f;
    }
};

int main() {
    int bad = 0;
    {
        Bowl F;
        double x[6] = {0, 0, 0, 0, 0, 0}, f = -1;
        int inner_total = 0;
        for (int outer = 0; outer < 6; ++outer) inner_total += wm::bfgs_minimize(F, x, 20, &f);  // (as GICP's outer loop restarts it)
        double err = 0;
        for (int i = 0; i < 6; ++i) err = std::fmax(err, std::fabs(x[i] - F.c[i]));
        std::printf("bfgs: %d inner iterations, %d evaluations, f %.3e, max |x - x*| %.3e\n", inner_total, F.evals, f, err);
        bad += !(err < 2e-3 && f < 1e-4 && F.evals < 400);
    }
    {
        double A[36], xs[6] = {1.5, -2.0, 0.25, 3.0, -0.75, 0.5}, b[6], x[6];
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) A[i * 6 + j] = (i == j ? 4.0 + i : 0.0) + 0.3 * std::sin(1.0 + i * 6 + j) + 0.3 * std::sin(1.0 + j * 6 + i);
        for (int i = 0; i < 6; ++i) {
            b[i] = 0;
            for (int j = 0; j < 6; ++j) b[i] += A[i * 6 + j] * xs[j];
        }
        wm::svd_solve6(A, b, x);
        double err = 0;
        for (int i = 0; i < 6; ++i) err = std::fmax(err, std::fabs(x[i] - xs[i]));
        std::printf("svd_solve6: max |x - x*| %.3e\n", err);
        bad += !(err < 1e-12);
        // a singular system: the minimum-norm solution of the consistent part (JacobiSVD::solve's behaviour)
        for (int j = 0; j < 6; ++j) A[5 * 6 + j] = A[4 * 6 + j], A[j * 6 + 5] = A[j * 6 + 4];
        A[35] = A[28];
        for (int i = 0; i < 6; ++i) {
            b[i] = 0;
            for (int j = 0; j < 6; ++j) b[i] += A[i * 6 + j] * xs[j];
        }
        wm::svd_solve6(A, b, x);
        double res = 0;
        for (int i = 0; i < 6; ++i) {
            double r = -b[i];
            for (int j = 0; j < 6; ++j) r += A[i * 6 + j] * x[j];
            res = std::fmax(res, std::fabs(r));
        }
        std::printf("svd_solve6 (rank 5): residual %.3e, x4 - x5 %.3e\n", res, x[4] - x[5]);
        bad += !(res < 1e-10 && std::fabs(x[4] - x[5]) < 1e-10);
    }
    std::printf("failed checks: %d\n", bad);
    return bad ? 1 : 0;
}
