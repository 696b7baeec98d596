// Developer probe: rocPRIM's onesweep radix sort of (u32 key, u32 value) pairs over the low `bits` bits, under different
// configurations (digit width, items per thread) -- what the Morton order of a cloud and the NDT voxel sort cost.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/dev/sort_probe scripts/dev/sort_probe.hip
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>
#include "../../libwave_amd/csrc/wm_sort.hpp"
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>

static float run_own(size_t n, unsigned bits, unsigned *k1, unsigned *k2, unsigned *v1, unsigned *v2, const std::vector<unsigned> &hk,
                     const std::vector<unsigned> &hv, const std::vector<unsigned> &expect) {
    void *d_tmp;
    (void) hipMalloc(&d_tmp, wm::rs_temp_bytes(n));
    hipEvent_t a, b;
    (void) hipEventCreate(&a);
    (void) hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 12; ++rep) {
        (void) hipMemcpy(k1, hk.data(), n * 4, hipMemcpyHostToDevice);
        (void) hipMemcpy(v1, hv.data(), n * 4, hipMemcpyHostToDevice);
        (void) hipEventRecord(a, 0);
        if (wm::rs_sort_pairs(d_tmp, k1, k2, v1, v2, n, bits, 0) != hipSuccess) { printf("own: sort failed\n"); return -1; }
        (void) hipEventRecord(b, 0);
        (void) hipEventSynchronize(b);
        float ms;
        (void) hipEventElapsedTime(&ms, a, b);
        best = std::min(best, ms);
    }
    std::vector<unsigned> got(n), gk(n);
    (void) hipMemcpy(got.data(), v2, n * 4, hipMemcpyDeviceToHost);
    (void) hipMemcpy(gk.data(), k2, n * 4, hipMemcpyDeviceToHost);
    bool ok = got == expect;
    for (size_t i = 0; ok && i < n; ++i) ok = gk[i] == hk[expect[i]];
    printf("%-44s n %zu bits %u: %7.1f us  %s\n", "own (hist + scan + scatter per pass)", n, bits, best * 1e3f, ok ? "stable order OK" : "WRONG ORDER");
    (void) hipFree(d_tmp);
    return best;
}

template <class Config>
static float run(const char *name, size_t n, unsigned bits, unsigned *k1, unsigned *k2, unsigned *v1, unsigned *v2, const std::vector<unsigned> &hk,
                 const std::vector<unsigned> &expect) {
    size_t tmp = 0;
    if (rocprim::radix_sort_pairs<Config>(nullptr, tmp, k1, k2, v1, v2, n, 0u, bits, 0) != hipSuccess) { printf("%s: size query failed\n", name); return -1; }
    void *d_tmp;
    (void) hipMalloc(&d_tmp, tmp);
    hipEvent_t a, b;
    (void) hipEventCreate(&a);
    (void) hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 12; ++rep) {
        (void) hipMemcpy(k1, hk.data(), n * 4, hipMemcpyHostToDevice);
        (void) hipEventRecord(a, 0);
        if (rocprim::radix_sort_pairs<Config>(d_tmp, tmp, k1, k2, v1, v2, n, 0u, bits, 0) != hipSuccess) { printf("%s: sort failed\n", name); return -1; }
        (void) hipEventRecord(b, 0);
        (void) hipEventSynchronize(b);
        float ms;
        (void) hipEventElapsedTime(&ms, a, b);
        best = std::min(best, ms);
    }
    std::vector<unsigned> got(n);
    (void) hipMemcpy(got.data(), v2, n * 4, hipMemcpyDeviceToHost);
    const bool ok = got == expect;
    printf("%-44s n %zu bits %u: %7.1f us  %s\n", name, n, bits, best * 1e3f, ok ? "stable order OK" : "WRONG ORDER");
    (void) hipFree(d_tmp);
    return best;
}

int main() {
    for (size_t n : {(size_t) 500000, (size_t) 1000000, (size_t) 2000000}) {
        const unsigned bits = 22;
        std::vector<unsigned> hk(n), hv(n), expect(n);
        std::mt19937 rng(7);
        for (size_t i = 0; i < n; ++i) hk[i] = rng() & ((1u << bits) - 1u), hv[i] = (unsigned) i;
        for (size_t i = 0; i < n; ++i) expect[i] = (unsigned) i;
        std::stable_sort(expect.begin(), expect.end(), [&](unsigned x, unsigned y) { return hk[x] < hk[y]; });
        std::vector<unsigned> expect16(n);
        for (size_t i = 0; i < n; ++i) expect16[i] = (unsigned) i;
        std::stable_sort(expect16.begin(), expect16.end(), [&](unsigned x, unsigned y) { return (hk[x] & 0xFFFFu) < (hk[y] & 0xFFFFu); });
        unsigned *k1, *k2, *v1, *v2;
        (void) hipMalloc(&k1, n * 4), (void) hipMalloc(&k2, n * 4), (void) hipMalloc(&v1, n * 4), (void) hipMalloc(&v2, n * 4);
        (void) hipMemcpy(v1, hv.data(), n * 4, hipMemcpyHostToDevice);
        using namespace rocprim;
        using dflt = radix_sort_config<default_config, default_config, default_config, 0>;
        run<dflt>("default onesweep", n, bits, k1, k2, v1, v2, hk, expect);
        run_own(n, bits, k1, k2, v1, v2, hk, hv, expect);
        run_own(n, 16, k1, k2, v1, v2, hk, hv, expect16);
        (void) hipMemcpy(v1, hv.data(), n * 4, hipMemcpyHostToDevice);
        using o8_12 = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<256, 12>, kernel_config<256, 12>, 8>, 0>;
        run<o8_12>("8 bits, 256 x 12", n, bits, k1, k2, v1, v2, hk, expect);
        using o8_16 = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<256, 16>, kernel_config<256, 16>, 8>, 0>;
        run<o8_16>("8 bits, 256 x 16", n, bits, k1, k2, v1, v2, hk, expect);
        using o8_8 = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<256, 12>, kernel_config<256, 8>, 8>, 0>;
        run<o8_8>("8 bits, 256 x 8", n, bits, k1, k2, v1, v2, hk, expect);
        using o8_m = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<256, 12>, kernel_config<256, 12>, 8, block_radix_rank_algorithm::match>, 0>;
        run<o8_m>("8 bits, 256 x 12, match", n, bits, k1, k2, v1, v2, hk, expect);
        using o8_m16 = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<256, 16>, kernel_config<256, 16>, 8, block_radix_rank_algorithm::match>, 0>;
        run<o8_m16>("8 bits, 256 x 16, match", n, bits, k1, k2, v1, v2, hk, expect);
        using o8_m24 = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<256, 16>, kernel_config<256, 24>, 8, block_radix_rank_algorithm::match>, 0>;
        run<o8_m24>("8 bits, 256 x 24, match", n, bits, k1, k2, v1, v2, hk, expect);
        using o8_b = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<256, 12>, kernel_config<256, 12>, 8, block_radix_rank_algorithm::basic_memoize>, 0>;
        run<o8_b>("8 bits, 256 x 12, basic_memoize", n, bits, k1, k2, v1, v2, hk, expect);
        using o6 = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<256, 12>, kernel_config<256, 12>, 6>, 0>;
        run<o6>("6 bits, 256 x 12 (4 passes)", n, bits, k1, k2, v1, v2, hk, expect);
        (void) hipFree(k1), (void) hipFree(k2), (void) hipFree(v1), (void) hipFree(v2);
    }
    return 0;
}
