// FETCH_SIZE calibration probes (scripts/gpu_fetch_calib.sh): what does rocprofv3's FETCH_SIZE count, per
// byte really requested from beyond the L2, for (a) a wide coalesced stream, (b) a gather of one 16-byte
// element per 128-byte line, (c) per 64-byte sector, (d) 64 contiguous bytes (four float4: one trip of the
// search's candidate walk) per 128-byte line?  The array (2 GiB) is far larger than L2 + Infinity Cache and
// every line / sector is touched exactly once per launch (index = i * odd constant mod 2^k: a bijection),
// so the bytes that MUST come from memory are known: lines x granule.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(256) k_stream(const float4 *a, size_t n, float *out) {
    float s = 0.f;
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) {
        const float4 v = a[i];
        s += v.x + v.w;
    }
    if (s == 123.456f) out[0] = s;
}

// one access of `vec` consecutive float4 at element index perm(i) * stride_elems
template <int VEC>
__global__ void __launch_bounds__(256) k_gather(const float4 *a, unsigned long long units, unsigned mask_bits,
                                                unsigned stride_elems, float *out) {
    float s = 0.f;
    const unsigned long long mask = (1ull << mask_bits) - 1ull;
    for (unsigned long long i = (unsigned long long) blockIdx.x * 256 + threadIdx.x; i < units;
         i += (unsigned long long) gridDim.x * 256) {
        const unsigned long long p = (i * 0x9E3779B97F4A7C15ull) & mask;  // odd multiplier: a permutation of [0, 2^k)
        const float4 *q = a + p * stride_elems;
#pragma unroll
        for (int u = 0; u < VEC; ++u) {
            const float4 v = q[u];
            s += v.x + v.w;
        }
    }
    if (s == 123.456f) out[0] = s;
}

int main() {
    const size_t bytes = 2ull << 30;
    float4 *a;
    float *out;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    hipMemset(a, 0, bytes);
    hipDeviceSynchronize();
    const size_t n = bytes / 16;
    const unsigned lines_bits = 24, sect_bits = 25;  // 2^24 lines of 128 B, 2^25 sectors of 64 B = 2 GiB
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, a, n, out);
        hipLaunchKernelGGL(k_gather<1>, dim3(4096), dim3(256), 0, 0, a, 1ull << lines_bits, lines_bits, 8u, out);   // 16 B per 128-B line
        hipLaunchKernelGGL(k_gather<1>, dim3(4096), dim3(256), 0, 0, a, 1ull << sect_bits, sect_bits, 4u, out);     // 16 B per 64-B sector
        hipLaunchKernelGGL(k_gather<4>, dim3(4096), dim3(256), 0, 0, a, 1ull << lines_bits, lines_bits, 8u, out);   // 64 B per 128-B line
        hipDeviceSynchronize();
    }
    std::printf("stream_bytes %zu lines %llu sectors %llu\n", bytes, 1ull << lines_bits, 1ull << sect_bits);
    return 0;
}
