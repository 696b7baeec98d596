// gicp_quad_eval (libwave_amd/csrc/wm_gicp_quad.hpp) compiled for the HOST: the evaluator of GICP's statistics objective
// is one source for the host path (wm_gicp.hip), the device (wm_gicp_small.hip) and -- restated in C -- the oracle.
// Reads "Q[74] T0[12] n" then n states x[6] (C99 hex floats) from stdin, prints f and the gradient of each state as hex
// floats; tests/test_gicp_quad_host_cpu.py compares them BIT FOR BIT with the oracle's evaluation of the same statistics.
#include <cstdio>

#include "wm_gicp_quad.hpp"

int main() {
    double Q[wm::kQuadN];
    float T0[12];
    for (int k = 0; k < wm::kQuadN; ++k)
        if (std::scanf("%la", &Q[k]) != 1) return 2;
    for (int k = 0; k < 12; ++k) {
        double v;
        if (std::scanf("%la", &v) != 1) return 2;
        T0[k] = (float) v;
    }
    int n = 0;
    if (std::scanf("%d", &n) != 1) return 2;
    double base[16];
    wm::mat4_identity(base);
    for (int i = 0; i < n; ++i) {
        double x[6], g[6];
        for (int k = 0; k < 6; ++k)
            if (std::scanf("%la", &x[k]) != 1) return 2;
        const double f = wm::gicp_quad_eval(Q, T0, base, x, g);
        std::printf("%a", f);
        for (int k = 0; k < 6; ++k) std::printf(" %a", g[k]);
        std::printf("\n");
    }
    return 0;
}
