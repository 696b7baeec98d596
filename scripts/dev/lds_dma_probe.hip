// Probe (developer): where do the lanes' bytes of global_load_lds_dwordx4 / dword land in LDS on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef const __attribute__((address_space(1))) void *gptr;
typedef __attribute__((address_space(3))) void *lptr;
__global__ void k16(const float4 *src, float4 *out) {
    __shared__ float4 buf[64];
    const unsigned lane = threadIdx.x;
    buf[lane] = make_float4(-1.f, -1.f, -1.f, -1.f);
    __syncthreads();
    __builtin_amdgcn_global_load_lds((gptr) (src + (63 - lane)), (lptr) buf, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[lane] = buf[lane];
}
__global__ void k16half(const float4 *src, float4 *out) {  // lanes < 32 only
    __shared__ float4 buf[64];
    const unsigned lane = threadIdx.x;
    buf[lane] = make_float4(-1.f, -1.f, -1.f, -1.f);
    __syncthreads();
    if (lane < 32) __builtin_amdgcn_global_load_lds((gptr) (src + lane), (lptr) buf, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[lane] = buf[lane];
}
int main() {
    std::vector<float4> h(64);
    for (int i = 0; i < 64; ++i) h[i] = make_float4((float) i, 100.f + i, 200.f + i, 300.f + i);
    float4 *d, *o;
    hipMalloc(&d, 64 * 16);
    hipMalloc(&o, 64 * 16);
    hipMemcpy(d, h.data(), 64 * 16, hipMemcpyHostToDevice);
    std::vector<float4> r(64);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(r.data(), o, 64 * 16, hipMemcpyDeviceToHost);
    printf("dwordx4, lane l loads src[63-l]: buf[0..3].x = %g %g %g %g ; buf[63].x = %g (expect 63 62 61 60 ; 0 if lane-indexed)\n", r[0].x, r[1].x, r[2].x, r[3].x, r[63].x);
    printf("   buf[0] = %g %g %g %g\n", r[0].x, r[0].y, r[0].z, r[0].w);
    hipLaunchKernelGGL(k16half, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(r.data(), o, 64 * 16, hipMemcpyDeviceToHost);
    printf("half exec: buf[0].x %g buf[31].x %g buf[32].x %g (expect 0 31 -1)\n", r[0].x, r[31].x, r[32].x);
    return 0;
}
