// wm_bins.hpp -- the iteration's sums as exact, order-independent integer accumulators (round 6).
//
// Up to round 5 every wave of a search kernel stored a row of 18 partial sums (doubles) and a reduction kernel
// (k_reduce_rows: 15 625 rows -> 123) plus the solve kernel's own first phase added them in a fixed order: two dependent
// launches behind every full search, a thousand-row sum in front of every solve behind a certified one.  Folding those
// into the producing kernel by a last-finisher ticket was measured this round and buys nothing (profiles/
// r06_experiments.md: the ticket chain costs what the launch did).  What does: NOT needing an order at all.
//
// A wave's partial sum x (a double) is cut into three signed 40-bit limbs of the fixed-point number trunc(x * 2^56)
//     x ~ l2 * 2^24 + l1 * 2^-16 + l0 * 2^-56,      |l_k| < 2^40        (|x| < 2^63; exact for |x| >= 2^-4)
// and every limb is ADDED, with a fire-and-forget 64-bit integer atomic at agent scope, into one of kBinCount bins
// (bin = the wave's row number mod kBinCount: spreads the traffic over memory channels).  Integer addition is associative
// and commutative: the bins' totals are the exact sum of the waves' limbs WHATEVER the order the atomics land in --
// bit-reproducible without a fixed order, up to 2^23 addends per bin limb.  The solve kernel (k_bins_solve, wm_icp.hip)
// adds the 64 bins (integers: any tree), turns the three totals back into a double by an error-free cascade, and
// puts the zeros back for the next iteration.  Measured (scripts/dev/atomic_probe.hip, 15 632 one-wave workgroups all
// reaching their atomics at once, the worst case): 10.2 us against 5.1 us for a row store each; with the waves' ends
// spread over a search kernel's 40-500 us the atomics cost nothing measurable.
//
// (the per-wave sums themselves are formed as before: f64 terms, recursive halving in a fixed lane order.  What changes
// is the sum ACROSS waves: exact instead of rounded after every addition -- a difference of 1e-16 relative.)
#ifndef WM_BINS_HPP
#define WM_BINS_HPP

#include "wm_internal.hpp"

namespace wm {

constexpr int kBinCount = 64;   // bins a launch's waves spread their additions over
constexpr int kBinLimbs = 3;
constexpr int kBinComps = kAcc + 1;   // the kAcc sums + [kAcc]: queries a certificate launch had to search
constexpr int kBinStride = 32;  // words per limb row of a bin (kBinComps used: a limb row is one 256-byte stretch)
constexpr size_t kBinWords = (size_t) kBinCount * kBinLimbs * kBinStride;

// an integer-valued double |v| < 2^51 -> int64 (the 1.5 * 2^52 trick: the sum's low mantissa bits ARE v + 2^51)
// (host + device: the splitting is checked on the CPU too, wm_debug_bins_sum / tests/test_bins_cpu.py)
__device__ __host__ inline long long bins_to_i64(double v) {
    const double d = v + 6755399441055744.0;  // 2^52 + 2^51
    long long bits;
    __builtin_memcpy(&bits, &d, sizeof(bits));
    return (bits & 0x000FFFFFFFFFFFFFll) - 0x0008000000000000ll;
}

__device__ __host__ inline void bins_split(double x, long long (&l)[kBinLimbs]) {
    const double t2 = trunc(x * 5.9604644775390625e-08);         // x * 2^-24
    const double r1 = fma(-t2, 16777216.0, x);                   // x - t2 * 2^24     (exact)
    const double t1 = trunc(r1 * 65536.0);                       // r1 * 2^16
    const double r0 = fma(-t1, 1.52587890625e-05, r1);           // r1 - t1 * 2^-16   (exact)
    const double t0 = trunc(r0 * 72057594037927936.0);           // r0 * 2^56
    l[2] = bins_to_i64(t2);
    l[1] = bins_to_i64(t1);
    l[0] = bins_to_i64(t0);
}

// component `comp` of bin `bin` += x  (three fire-and-forget atomics; the caller does not wait for them)
// (a sum the format cannot hold -- not finite, or 2^62 and beyond: coordinates of ~1e7 m on millions of points --
// raises the bin's POISON word instead, and the solve ends the registration with "no correspondences": loud, not wrong)
constexpr unsigned kBinPoison = kBinStride - 1;  // word of limb row 0
__device__ __forceinline__ void bins_add(long long *bins, unsigned bin, unsigned comp, double x) {
    long long l[kBinLimbs];
    long long *p = bins + (size_t) bin * (kBinLimbs * kBinStride) + comp;
    if (!(fabs(x) < 4611686018427387904.0)) {
        (void) __hip_atomic_fetch_add(bins + (size_t) bin * (kBinLimbs * kBinStride) + kBinPoison, 1ll, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    bins_split(x, l);
#pragma unroll
    for (int k = 0; k < kBinLimbs; ++k)
        (void) __hip_atomic_fetch_add(p + k * kBinStride, l[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ... an integer count (the searched queries): limb 0 only, unscaled
__device__ __forceinline__ void bins_add_count(long long *bins, unsigned bin, unsigned comp, long long n) {
    (void) __hip_atomic_fetch_add(bins + (size_t) bin * (kBinLimbs * kBinStride) + comp, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the double nearest (to ~1e-32 relative) to L2 * 2^24 + L1 * 2^-16 + L0 * 2^-56 for int64 limb totals  (bins_value, below)
__device__ __host__ inline double bins_value(long long L0, long long L1, long long L2);

// ---- the reading side: the bins' words added up (integers: exact in any order) and set back to zero, by ALL THREADS
// threads of ONE workgroup (THREADS >= 4 x 57); afterwards -- a barrier has been passed -- B.tot[c] is component c as a
// double ([kAcc]: the searched-queries count) and B.poison says whether a sum did not fit the limbs.
struct BinsLds {
    long long part[4][kBinLimbs * kBinComps];
    double tot[kBinComps];
    unsigned poison;
};
template <int THREADS>
__device__ __forceinline__ void bins_collect(long long *__restrict__ bins, BinsLds &B) {
    constexpr int kWordsPerBin = kBinLimbs * kBinComps;  // 57 words of a bin are in use
    constexpr int kGroups = 4;                            // groups of threads, kBinCount / 4 bins each
    static_assert(THREADS >= kGroups * kWordsPerBin && kBinCount % kGroups == 0, "bins per thread group");
    constexpr int kPer = kBinCount / kGroups;
    const unsigned g = threadIdx.x / (unsigned) kWordsPerBin, j = threadIdx.x % (unsigned) kWordsPerBin;
    const unsigned limb = j / (unsigned) kBinComps, comp = j % (unsigned) kBinComps;
    if (threadIdx.x == 0) B.poison = 0u;
    if (g < (unsigned) kGroups) {
        long long v[kPer];
#pragma unroll
        for (int b = 0; b < kPer; ++b)  // (sixteen loads in flight: one round trip)
            v[b] = bins[((size_t) (g * kPer + b) * kBinLimbs + limb) * kBinStride + comp];
        long long t = 0;
#pragma unroll
        for (int b = 0; b < kPer; ++b) t += v[b];
        B.part[g][j] = t;
#pragma unroll
        for (int b = 0; b < kPer; ++b)  // zeros for the next iteration
            bins[((size_t) (g * kPer + b) * kBinLimbs + limb) * kBinStride + comp] = 0ll;
    }
    __syncthreads();
    if (threadIdx.x < (unsigned) kBinCount) {  // a sum that the limbs could not hold?
        long long *pw = bins + (size_t) threadIdx.x * (kBinLimbs * kBinStride) + kBinPoison;
        if (*pw != 0ll) {
            atomicOr(&B.poison, 1u);
            *pw = 0ll;
        }
    }
    if (threadIdx.x < (unsigned) kBinComps) {
        long long L[kBinLimbs];
#pragma unroll
        for (int l = 0; l < kBinLimbs; ++l) {
            long long t = 0;
#pragma unroll
            for (int gg = 0; gg < kGroups; ++gg) t += B.part[gg][l * kBinComps + threadIdx.x];
            L[l] = t;
        }
        // ([kAcc]: the searched-queries count, added unscaled into limb 0)
        B.tot[threadIdx.x] = threadIdx.x < (unsigned) kAcc ? bins_value(L[0], L[1], L[2]) : (double) L[0];
    }
    __syncthreads();
}

// the double nearest (to ~1e-32 relative) to L2 * 2^24 + L1 * 2^-16 + L0 * 2^-56 for int64 limb totals: every limb total
// is cut into two halves that a double holds exactly, the six terms are added largest first with an error-free TwoSum
// cascade.  The same operations on the same integers give the same double on every device and on the host.
__device__ __host__ inline double bins_value(long long L0, long long L1, long long L2) {
    double term[6];
    const long long L[3] = {L2, L1, L0};
    const double scale[3] = {16777216.0, 1.52587890625e-05, 1.3877787807814457e-17};  // 2^24, 2^-16, 2^-56
    for (int k = 0; k < 3; ++k) {
        const long long hi = L[k] >> 32;                        // (arithmetic shift: floor)
        const long long lo = L[k] - hi * 4294967296ll;          // 0 .. 2^32 - 1
        term[2 * k] = (double) hi * 4294967296.0 * scale[k];
        term[2 * k + 1] = (double) lo * scale[k];
    }
    double s = 0.0, c = 0.0;  // s + c = the running sum (c: what s lost)
    for (int k = 0; k < 6; ++k) {
        const double t = s + term[k];
        const double bb = t - s;
        const double err = (s - (t - bb)) + (term[k] - bb);   // TwoSum
        s = t;
        c += err;
    }
    return s + c;
}

}  // namespace wm

#endif  // WM_BINS_HPP
