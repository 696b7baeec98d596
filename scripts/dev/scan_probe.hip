// Developer probe: rocPRIM's exclusive scan of n unsigned counts (+ the total behind them, as wm_grid.hip's exclusive_scan
// asks for it) under different configurations.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/dev/scan_probe scripts/dev/scan_probe.hip
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <algorithm>

struct ScanIn {
    const unsigned *in;
    size_t n;
    __host__ __device__ unsigned operator()(size_t i) const { return i < n ? in[i] : 0u; }
};

template <class Config>
static void run(const char *name, const unsigned *in, unsigned *out, size_t n, const std::vector<unsigned> &expect_tail) {
    auto it = rocprim::make_transform_iterator(rocprim::counting_iterator<size_t>(0), ScanIn{in, n});
    size_t bytes = 0;
    if (rocprim::exclusive_scan<Config>(nullptr, bytes, it, out, 0u, n + 1, rocprim::plus<unsigned>(), 0) != hipSuccess) { printf("%s: query failed\n", name); return; }
    void *tmp;
    (void) hipMalloc(&tmp, bytes + 64);
    hipEvent_t a, b;
    (void) hipEventCreate(&a), (void) hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 12; ++rep) {
        (void) hipEventRecord(a, 0);
        (void) rocprim::exclusive_scan<Config>(tmp, bytes, it, out, 0u, n + 1, rocprim::plus<unsigned>(), 0);
        (void) hipEventRecord(b, 0);
        (void) hipEventSynchronize(b);
        float ms;
        (void) hipEventElapsedTime(&ms, a, b);
        best = std::min(best, ms);
    }
    unsigned total = 0;
    (void) hipMemcpy(&total, out + n, 4, hipMemcpyDeviceToHost);
    printf("%-46s n %zu: %7.1f us  total %u (%s)\n", name, n, best * 1e3f, total, total == expect_tail[0] ? "OK" : "WRONG");
    (void) hipFree(tmp);
}

int main() {
    for (size_t n : {(size_t) 14000000, (size_t) 1750000, (size_t) 2000000}) {
        std::vector<unsigned> h(n);
        unsigned long long tot = 0;
        for (size_t i = 0; i < n; ++i) h[i] = (unsigned) ((i * 2654435761u) >> 30) & 1u, tot += h[i];
        std::vector<unsigned> expect{(unsigned) tot};
        unsigned *in, *out;
        (void) hipMalloc(&in, n * 4), (void) hipMalloc(&out, (n + 1) * 4);
        (void) hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
        using namespace rocprim;
        run<default_config>("default", in, out, n, expect);
        run<scan_config<256, 16, block_load_method::block_load_transpose, block_store_method::block_store_transpose, block_scan_algorithm::using_warp_scan>>("256 x 16 transpose", in, out, n, expect);
        run<scan_config<256, 24, block_load_method::block_load_transpose, block_store_method::block_store_transpose, block_scan_algorithm::using_warp_scan>>("256 x 24 transpose", in, out, n, expect);
        run<scan_config<256, 32, block_load_method::block_load_transpose, block_store_method::block_store_transpose, block_scan_algorithm::using_warp_scan>>("256 x 32 transpose", in, out, n, expect);
        run<scan_config<128, 32, block_load_method::block_load_transpose, block_store_method::block_store_transpose, block_scan_algorithm::using_warp_scan>>("128 x 32 transpose", in, out, n, expect);
        run<scan_config<256, 16, block_load_method::block_load_direct, block_store_method::block_store_direct, block_scan_algorithm::using_warp_scan>>("256 x 16 direct", in, out, n, expect);
        run<scan_config<256, 16, block_load_method::block_load_transpose, block_store_method::block_store_transpose, block_scan_algorithm::reduce_then_scan>>("256 x 16 transpose, reduce_then_scan", in, out, n, expect);
        (void) hipFree(in), (void) hipFree(out);
    }
    return 0;
}
