// developer probe: device-to-device copy rate by kernel shape (what wm_debug_copy_bandwidth should use)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4v __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_copy(const f4v *__restrict__ a, f4v *__restrict__ b, size_t n) {
    const size_t stride = (size_t) gridDim.x * 256u;
    size_t i = (size_t) blockIdx.x * 256u + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4v v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], b + i + u * stride);
            else b[i + u * stride] = v[u];
        }
    }
    for (; i < n; i += stride) b[i] = a[i];
}
template <int U, bool NT>
static void run(const f4v *a, f4v *b, size_t n, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_copy<U, NT>), dim3(blocks), dim3(256), 0, 0, a, b, n);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k_copy<U, NT>), dim3(blocks), dim3(256), 0, 0, a, b, n);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("U=%d nt=%d blocks=%5d : %.0f GB/s\n", U, (int) NT, blocks, 2.0 * n * 16 * 10 / (ms * 1e-3) / 1e9);
}
int main() {
    const size_t n = (size_t) 1 << 26;  // 1 GiB
    f4v *a, *b;
    hipMalloc(&a, n * 16);
    hipMalloc(&b, n * 16);
    hipMemset(a, 1, n * 16);
    for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536}) {
        run<1, false>(a, b, n, blocks);
        run<4, false>(a, b, n, blocks);
        run<8, false>(a, b, n, blocks);
        run<4, true>(a, b, n, blocks);
    }
    return 0;
}
