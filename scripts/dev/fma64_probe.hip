// Developer probe: the dependent-issue latency of v_fma_f64 on gfx950 (one wave, a chain of N dependent FMAs),
// with all 64 lanes and with 12 lanes active, and of a chain fed from LDS (ds_read_b128 a chunk ahead).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fma64_probe scripts/dev/fma64_probe.hip && /tmp/fma64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_chain(double *out, long long *cyc, int n, int lanes, double x, double y) {
    double acc = out[threadIdx.x];
    long long t0 = 0, t1 = 0;
    if ((int) threadIdx.x < lanes) {
        t0 = __builtin_readcyclecounter();
#pragma unroll 16
        for (int i = 0; i < n; ++i) acc = fma(x, y, acc);
        t1 = __builtin_readcyclecounter();
    }
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_indep(double *out, long long *cyc, int n, double x, double y) {
    double a0 = out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 4
    for (int i = 0; i < n; i += 4) {
        a0 = fma(x, y, a0);
        a1 = fma(x, y, a1);
        a2 = fma(x, y, a2);
        a3 = fma(x, y, a3);
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double *out;
    long long *cyc;
    hipMalloc(&out, 64 * 8);
    hipMalloc(&cyc, 8);
    hipMemset(out, 0, 64 * 8);
    const int n = 1 << 16;
    for (int rep = 0; rep < 2; ++rep)
        for (int lanes : {64, 12, 1}) {
            hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, out, cyc, n, lanes, 1.0000001, 0.9999999);
            long long c;
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("dependent v_fma_f64, %2d lanes: %.2f ticks per FMA\n", lanes, (double) c / n);
        }
    hipLaunchKernelGGL(k_indep, dim3(1), dim3(64), 0, 0, out, cyc, n, 1.0000001, 0.9999999);
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("4 independent chains: %.2f ticks per FMA\n", (double) c / n);
    // wall clock of the chain, to turn ticks into time
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, out, cyc, n * 16, 12, 1.0000001, 0.9999999);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("chain of %d: %.3f ms = %.2f ns per FMA; %lld ticks = %.1f ticks/us\n", n * 16, ms, ms * 1e6 / (n * 16), c, c / (ms * 1e3));
    return 0;
}
