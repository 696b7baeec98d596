// What does it cost 15 625 short waves to ADD their 18 partial sums into shared bins with fire-and-forget integer
// atomics (three 40-bit limbs per sum: exact and order-independent), instead of storing a row each for a reduction
// kernel to add up?  (scripts/dev: a probe, not product code)
//   modes: 0 = a 144-byte row per wave (what k_nn_grid does today)
//          1 = agent-scope atomics (sc1: performed where every XCD sees them) into `nb` bins
//          2 = atomics performed in the issuing XCD's own L2 (no scope bits) into bins of that XCD (HW_REG_XCC_ID) x `nb`
//   usage: atomic_probe <mode> <nb> <spin> [waves]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kAcc = 18, kLimbs = 3;

__device__ __forceinline__ unsigned xcc_id() {
    // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4)
    return (unsigned) __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11));
}

template <int MODE>
__global__ void __launch_bounds__(64) k_probe(double *rows, long long *bins, unsigned nb, unsigned spin, unsigned *xcc_seen) {
    const unsigned lane = threadIdx.x;
    // some work first: a wave of the search kernel lives ~15 us
    float a = (float) lane, b = 1.0001f;
    for (unsigned k = 0; k < spin; ++k) a = a * b + 0.5f;
    const long long v = (long long) (blockIdx.x % 1000u) + (long long) lane + (a < 0.f ? 1 : 0);  // (the value each lane adds)
    if (MODE == 0) {
        if (lane < (unsigned) kAcc) rows[(size_t) blockIdx.x * kAcc + lane] = (double) v;
    } else {
        unsigned bin;
        if (MODE == 2) {
            const unsigned x = xcc_id() & 7u;
            if (lane == 0 && xcc_seen) xcc_seen[blockIdx.x] = x;
            bin = x * nb + (blockIdx.x / 8u) % nb;
        } else {
            bin = blockIdx.x % nb;
        }
        if (lane < (unsigned) kAcc) {
            long long *p = bins + ((size_t) bin * kLimbs) * 32 + lane;  // a limb's 18 words in one 256-byte stretch
#pragma unroll
            for (int l = 0; l < kLimbs; ++l) {
                const long long add = l == 0 ? v : (l == 1 ? 2 * v : -v);
                if (MODE == 1) (void) __hip_atomic_fetch_add(p + l * 32, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else (void) __hip_atomic_fetch_add(p + l * 32, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
}

int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const unsigned nb = argc > 2 ? (unsigned) atoi(argv[2]) : 8u;
    const unsigned spin = argc > 3 ? (unsigned) atoi(argv[3]) : 0u;
    const unsigned waves = argc > 4 ? (unsigned) atoi(argv[4]) : 15625u + 7u;
    double *rows;
    long long *bins;
    unsigned *xs;
    const size_t nbins = (size_t) 8 * nb;
    CK(hipMalloc((void **) &rows, (size_t) waves * kAcc * 8));
    CK(hipMalloc((void **) &bins, nbins * kLimbs * 32 * 8));
    CK(hipMalloc((void **) &xs, (size_t) waves * 4));
    CK(hipMemset(bins, 0, nbins * kLimbs * 32 * 8));
    CK(hipMemset(xs, 0xFF, (size_t) waves * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 40;
    auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(waves), dim3(64), 0, 0, rows, bins, nb, spin, xs);
        else if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(waves), dim3(64), 0, 0, rows, bins, nb, spin, xs);
        else hipLaunchKernelGGL(k_probe<2>, dim3(waves), dim3(64), 0, 0, rows, bins, nb, spin, xs);
    };
    for (int k = 0; k < 5; ++k) launch();
    CK(hipDeviceSynchronize());
    CK(hipMemset(bins, 0, nbins * kLimbs * 32 * 8));
    CK(hipEventRecord(e0, 0));
    for (int k = 0; k < reps; ++k) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    // check the sums (modes 1, 2)
    bool ok = true;
    if (mode != 0) {
        std::vector<long long> h(nbins * kLimbs * 32);
        CK(hipMemcpy(h.data(), bins, h.size() * 8, hipMemcpyDeviceToHost));
        for (int c = 0; c < kAcc && ok; ++c) {
            long long want = 0;
            for (unsigned w = 0; w < waves; ++w) want += (long long) (w % 1000u) + c;
            want *= reps;
            long long got[kLimbs] = {0, 0, 0};
            for (size_t b = 0; b < nbins; ++b)
                for (int l = 0; l < kLimbs; ++l) got[l] += h[(b * kLimbs + l) * 32 + c];
            if (got[0] != want || got[1] != 2 * want || got[2] != -want) {
                ok = false;
                printf("component %d: got %lld %lld %lld, want %lld\n", c, got[0], got[1], got[2], want);
            }
        }
    }
    unsigned hist[16] = {0};
    if (mode == 2) {
        std::vector<unsigned> hx(waves);
        CK(hipMemcpy(hx.data(), xs, (size_t) waves * 4, hipMemcpyDeviceToHost));
        unsigned agree = 0;
        for (unsigned w = 0; w < waves; ++w) {
            if (hx[w] < 16) hist[hx[w]]++;
            if (hx[w] == (w & 7u)) agree++;
        }
        printf("XCC_ID == blockIdx %% 8 for %u of %u workgroups; per XCC:", agree, waves);
        for (int k = 0; k < 8; ++k) printf(" %u", hist[k]);
        printf("\n");
    }
    printf("mode %d nb %u spin %u waves %u: %.2f us per launch, sums %s\n", mode, nb, spin, waves, ms / reps * 1e3, mode == 0 ? "n/a" : (ok ? "exact" : "WRONG"));
    return ok ? 0 : 2;
}
