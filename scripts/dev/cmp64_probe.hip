// Developer probe: issue cost of v_cmp_lt_u64 against v_cmp_lt_u32 (+ what the k-NN insertion is made of) on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/dev/cmp64_probe scripts/dev/cmp64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(unsigned long long *out, long long *cyc, int n) {
    unsigned long long a = out[threadIdx.x], b = out[threadIdx.x + 64];
    unsigned acc = 0;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            if (MODE == 0) {  // (the compare feeds a select: a VALU-only chain)
                asm volatile("v_cmp_lt_u64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(acc) : "v"(a), "v"(b), "v"((unsigned) b) : "vcc");
            } else if (MODE == 1) {
                asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(acc) : "v"((unsigned) a), "v"((unsigned) b), "v"((unsigned) b) : "vcc");
            } else if (MODE == 3) {
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc) : "v"((unsigned) b) : "vcc");
            } else if (MODE == 2) {
                unsigned r;
                asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"((unsigned) a), "v"((unsigned) b), "v"(acc));
                acc = r;
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    unsigned long long *out;
    long long *cyc;
    (void) hipMalloc(&out, 128 * 8);
    (void) hipMalloc(&cyc, 8);
    (void) hipMemset(out, 1, 128 * 8);
    const int n = 4096;
    long long c;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, cyc, n);
        (void) hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("v_cmp_lt_u64 + v_cndmask: %.2f ticks per pair\n", (double) c / (n * 32));
        hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out, cyc, n);
        (void) hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("v_cmp_lt_u32 + v_cndmask: %.2f ticks per pair\n", (double) c / (n * 32));
        hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, out, cyc, n);
        (void) hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("v_cndmask alone (dependent): %.2f ticks each\n", (double) c / (n * 32));
        hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, out, cyc, n);
        (void) hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("v_med3_u32 (dependent): %.2f ticks each\n", (double) c / (n * 32));
    }
    return 0;
}
