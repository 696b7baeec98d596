// libm_sincosf (libwave_amd/csrc/wm_bfgs.hpp: what the batched GICP kernel builds its float transform from) against
// the installed libm's sinf / cosf, which the one-pair path and PCL call: the same float on (nearly) every argument.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "wm_bfgs.hpp"

int main(int argc, char **argv) {
    const long n = argc > 1 ? std::atol(argv[1]) : 6000000;
    std::srand(2);
    long bad_s = 0, bad_c = 0;
    for (long i = 0; i < n; ++i) {
        float x = ((float) std::rand() / (float) RAND_MAX - 0.5f) * ((i % 3) == 0 ? 0.2f : (i % 3) == 1 ? 6.5f : 230.f);
        if (i % 1000 == 7) x *= 1e-4f;
        if (i % 1000 == 8) x = (i & 1024) ? 0.75f : std::nextafterf(0.75f, 0.f);
        bad_s += wm::libm_sincosf(x, 0) != sinf(x);
        bad_c += wm::libm_sincosf(x, 1) != cosf(x);
    }
    long bad_a = 0, bad_as = 0;
    for (long i = 0; i < n; ++i) {
        const float a = ((float) std::rand() / (float) RAND_MAX - 0.5f) * ((i & 3) == 0 ? 2.0f : (i & 3) == 1 ? 20.f : (i & 3) == 2 ? 0.1f : ((i & 8) ? 1e4f : 1e9f));
        const float b = ((float) std::rand() / (float) RAND_MAX - 0.5f) * ((i & 4) ? 2.0f : 0.3f);
        const float s = ((float) std::rand() / (float) RAND_MAX - 0.5f) * ((i & 16) ? 2.0f : 0.1f);
        bad_a += wm::libm_atan2f(a, b) != atan2f(a, b);
        bad_a += wm::libm_atanf(a) != atanf(a);
        bad_as += wm::libm_asinf(s) != asinf(s);
    }
    std::printf("arguments %ld sin_mismatch %ld cos_mismatch %ld atan2_mismatch %ld asin_mismatch %ld\n", n, bad_s, bad_c, bad_a, bad_as);
    return 0;
}
