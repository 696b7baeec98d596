// wm_sort.hpp -- the stable LSD radix sort of (key, value) pairs over the low `bits` bits of the key that the voxel grids,
// the Morton order of a cloud and the NDT voxel model are built on.
//
// Two implementations behind one call, both stable (the result does not depend on the choice):
//
// * rs_sort_pairs (round 6; clouds above `radix_min` items): eight bits per pass, three launches per pass and nothing
//   else -- per-tile digit histograms (k_rs_hist), the exclusive scan of the digit-major histogram table (k_rs_scan: a
//   workgroup per digit row), the scatter (k_rs_scatter: a tile of 4 096 items in sixteen rounds of 256, ranks within a
//   round by wave-wide matching of the digit, the tile digit-sorted in LDS, then written out in runs).  No memsets, no
//   temporary but the table.  22-bit keys, 0.5M / 1M / 2M pairs (scripts/dev/sort_probe.hip): 77 / 82 / 94 us against
//   rocPRIM's 101 / 107 / 118 -- whose onesweep path is a histogram kernel, a scan, and per pass two memsets of its
//   look-back state and a 26-33 us pass: twelve dependent launches whose floor is their number, not their bytes (nine
//   here; per pass at 1M: histogram 6.5, scan 4.8, scatter 13.8 us).
// * rocPRIM (small clouds: its merge sort; WM_TUNE_SORT=0: its onesweep radix sort as before).
//   rocPRIM's default dispatch takes its MERGE sort for up to 1 M items (log2(n / 4096) merge passes of two kernels each,
//   whatever the key width); the keys here are narrow (17-30 bits: leaf / Morton indices), so above `radix_min` items
//   (256 k by default) the radix path is selected explicitly (merge-sort limit 0).
#ifndef WM_SORT_HPP
#define WM_SORT_HPP

#include <rocprim/rocprim.hpp>

namespace wm {

constexpr int kRsThreads = 256;                    // threads of a scatter / histogram workgroup
constexpr int kRsRounds = 16;                      // items per thread
constexpr int kRsTile = kRsThreads * kRsRounds;    // items per tile
constexpr int kRsBins = 256;                       // eight bits per pass

// hist[d * tiles + t] = number of items of tile t whose digit is d
template <class K>
__global__ void __launch_bounds__(kRsThreads)
    k_rs_hist(const K *__restrict__ keys, unsigned n, unsigned shift, unsigned dmask, unsigned *__restrict__ hist, unsigned tiles) {
    __shared__ unsigned h[kRsBins];
    const unsigned tid = threadIdx.x, tile = blockIdx.x;
    h[tid] = 0u;
    __syncthreads();
    const unsigned base = tile * (unsigned) kRsTile;
#pragma unroll
    for (int r = 0; r < kRsRounds; ++r) {
        const unsigned i = base + (unsigned) r * kRsThreads + tid;
        if (i < n) atomicAdd(&h[(unsigned) (keys[i] >> shift) & dmask], 1u);
    }
    __syncthreads();
    hist[(size_t) tid * tiles + tile] = h[tid];
}

// a barrier that waits for this wave's LDS traffic only (HIP's __syncthreads also waits for every outstanding global store:
// the scatter's rounds would each pay a trip to memory)
__device__ __forceinline__ void rs_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// exclusive sum over the 256 threads of a workgroup (thread t's value -> the sum of the values of threads < t)
__device__ __forceinline__ unsigned rs_block_exclusive(unsigned v, unsigned *s_wave /*[4]*/) {
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(incl, off);
        if (lane >= (unsigned) off) incl += t;
    }
    if (lane == 63u) s_wave[wave] = incl;
    __syncthreads();
    unsigned run = incl - v;
    for (unsigned w = 0; w < wave; ++w) run += s_wave[w];
    return run;
}

// row d of the table (digit d's counts, tile by tile) -> its exclusive prefix in place, its total -> totals[d]: one
// workgroup per digit, coalesced
static __global__ void __launch_bounds__(kRsThreads) k_rs_scan(unsigned *__restrict__ hist, unsigned tiles, unsigned *__restrict__ totals) {
    __shared__ unsigned s_wave[kRsThreads / 64];
    __shared__ unsigned s_carry;
    unsigned *row = hist + (size_t) blockIdx.x * tiles;
    if (threadIdx.x == 0) s_carry = 0u;
    __syncthreads();
    for (unsigned t0 = 0; t0 < tiles; t0 += kRsThreads) {
        const unsigned t = t0 + threadIdx.x;
        const unsigned v = t < tiles ? row[t] : 0u;
        const unsigned ex = rs_block_exclusive(v, s_wave);
        const unsigned carry = s_carry;
        if (t < tiles) row[t] = carry + ex;
        __syncthreads();  // (everybody has read s_carry and s_wave)
        if (threadIdx.x == kRsThreads - 1) s_carry = carry + ex + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = s_carry;
}

// The items of a tile to their places.  Round r holds items base + r * 256 + tid (index order = round, then thread); a
// wave finds the lanes that share its lane's digit (eight ballots), a round's four waves count their digits into LDS, the
// digits' running offsets move on behind every round -- the same item order within a digit as the input's.  The places
// are first places WITHIN THE TILE (digit-sorted, in LDS); the tile then leaves in that order: the items of a digit are
// neighbours in memory (64-byte runs on average), where a store per item from the rounds was a partial line each (the
// whole sort of 2M pairs: 113 -> 94 us; at 1M and below the extra trip through LDS costs 7 us).
template <class K, class V>
__global__ void __launch_bounds__(kRsThreads)
    k_rs_scatter(const K *__restrict__ keys_in, const V *__restrict__ vals_in, K *__restrict__ keys_out,
                 V *__restrict__ vals_out, unsigned n, unsigned shift, unsigned dmask, const unsigned *__restrict__ hist,
                 unsigned tiles, const unsigned *__restrict__ totals) {
    constexpr int kWaves = kRsThreads / 64;
    __shared__ unsigned s_gbase[kRsBins];   // digit d's first place in the output, for this tile
    __shared__ unsigned s_lstart[kRsBins];  // ... within the tile
    __shared__ unsigned s_run[kRsBins];     // ... the next free place within the tile
    __shared__ unsigned s_cnt[kWaves][kRsBins];
    __shared__ unsigned s_wave[kWaves];
    __shared__ K s_key[kRsTile];
    __shared__ V s_val[kRsTile];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, tile = blockIdx.x;
    const unsigned base = tile * (unsigned) kRsTile;
    const unsigned count = min((unsigned) kRsTile, n - base);
    K key[kRsRounds];
    V val[kRsRounds];
#pragma unroll
    for (int r = 0; r < kRsRounds; ++r) {  // (all sixteen rounds' loads in flight before the first is used)
        const unsigned i = base + (unsigned) r * kRsThreads + tid;
        key[r] = i < n ? keys_in[i] : (K) 0;
        val[r] = i < n ? vals_in[i] : (V) 0;
    }
    // digit t's first place = the items of every smaller digit + digit t's items of the tiles before this one
    const unsigned before = rs_block_exclusive(totals[tid], s_wave);
    s_gbase[tid] = before + hist[(size_t) tid * tiles + tile];
    s_run[tid] = 0u;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) s_cnt[w][tid] = 0u;
    __syncthreads();
    // the tile's own digit counts -> the digits' first places within the tile
#pragma unroll
    for (int r = 0; r < kRsRounds; ++r) {
        const unsigned i = base + (unsigned) r * kRsThreads + tid;
        if (i < n) atomicAdd(&s_run[(unsigned) (key[r] >> shift) & dmask], 1u);
    }
    __syncthreads();
    {
        const unsigned mine = s_run[tid];
        const unsigned ex = rs_block_exclusive(mine, s_wave);
        __syncthreads();
        s_lstart[tid] = ex;
        s_run[tid] = ex;
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < kRsRounds; ++r) {  // (unrolled: key[r] / val[r] stay in registers)
        const unsigned i = base + (unsigned) r * kRsThreads + tid;
        const bool live = i < n;
        const unsigned d = (unsigned) (key[r] >> shift) & dmask;
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const bool one = ((d >> bit) & 1u) != 0u;
            const unsigned long long m = __ballot(one);
            peers &= one ? m : ~m;
        }
        const unsigned rank = (unsigned) __popcll(peers & lt);
        if (live && rank == 0u) s_cnt[wave][d] = (unsigned) __popcll(peers);
        rs_lds_barrier();
        if (live) {
            unsigned pos = s_run[d] + rank;
            for (unsigned w = 0; w < wave; ++w) pos += s_cnt[w][d];
            s_key[pos] = key[r];
            s_val[pos] = val[r];
        }
        rs_lds_barrier();
        {  // (thread t looks after digit t)
            unsigned add = 0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) {
                add += s_cnt[w][tid];
                s_cnt[w][tid] = 0u;
            }
            s_run[tid] += add;
        }
        rs_lds_barrier();
    }
    // the tile leaves in digit order
#pragma unroll 4
    for (unsigned j = tid; j < count; j += kRsThreads) {
        const K k = s_key[j];
        const unsigned d = (unsigned) (k >> shift) & dmask;
        const unsigned pos = s_gbase[d] + (j - s_lstart[d]);
        keys_out[pos] = k;
        vals_out[pos] = s_val[j];
    }
}

// bytes of temporary storage rs_sort_pairs needs for n items
inline size_t rs_temp_bytes(size_t n) {
    const size_t tiles = (n + kRsTile - 1) / kRsTile;
    return (tiles + 1) * kRsBins * sizeof(unsigned) + 256;  // the table + the digits' totals
}

// keys_out / vals_out = the pairs in ascending order of the low `bits` bits of the key, stable; keys_in / vals_in are
// overwritten (ping-pong buffers)
template <class K, class V>
inline hipError_t rs_sort_pairs(void *tmp, K *keys_in, K *keys_out, V *vals_in, V *vals_out, size_t n, unsigned bits,
                                hipStream_t stream) {
    const unsigned tiles = (unsigned) ((n + kRsTile - 1) / kRsTile);
    unsigned *hist = static_cast<unsigned *>(tmp);
    unsigned *totals = hist + (size_t) tiles * kRsBins;
    const unsigned passes = bits == 0 ? 1u : (bits + 7u) / 8u;
    K *ki = keys_in, *ko = keys_out;
    V *vi = vals_in, *vo = vals_out;
    if ((passes & 1u) == 0u) {  // an even number of passes ends where it started: start from the output buffers
        hipError_t e = hipMemcpyAsync(keys_out, keys_in, n * sizeof(K), hipMemcpyDeviceToDevice, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(vals_out, vals_in, n * sizeof(V), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
        ki = keys_out, ko = keys_in, vi = vals_out, vo = vals_in;
    }
    for (unsigned p = 0; p < passes; ++p) {
        const unsigned shift = 8u * p;
        // (the last pass of a key whose width is not a multiple of the digit looks at the bits that are left, no further)
        const unsigned left = bits > shift ? bits - shift : 0u;
        const unsigned dmask = left >= 8u ? (unsigned) kRsBins - 1u : (1u << left) - 1u;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rs_hist<K>), dim3(tiles), dim3(kRsThreads), 0, stream, (const K *) ki, (unsigned) n, shift, dmask, hist, tiles);
        hipLaunchKernelGGL(k_rs_scan, dim3(kRsBins), dim3(kRsThreads), 0, stream, hist, tiles, totals);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rs_scatter<K, V>), dim3(tiles), dim3(kRsThreads), 0, stream, (const K *) ki, (const V *) vi, ko, vo,
                           (unsigned) n, shift, dmask, (const unsigned *) hist, tiles, (const unsigned *) totals);
        K *tk = ki;
        ki = ko, ko = tk;
        V *tv = vi;
        vi = vo, vo = tv;
    }
    return hipGetLastError();
}

// own: 1 = rs_sort_pairs above radix_min items (the default), 0 = rocPRIM throughout
template <class K, class V>
inline hipError_t sort_pairs_low_bits(void *tmp, size_t &tmp_bytes, K *keys_in, K *keys_out, V *vals_in,
                                      V *vals_out, size_t n, unsigned bits, hipStream_t stream,
                                      size_t radix_min = (size_t) (256u << 10), int own = 1) {
    using radix_only = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                  rocprim::default_config, 0>;
    if (n > radix_min && own && n < (size_t) 0xFFFF0000u) {
        if (tmp == nullptr) {
            tmp_bytes = rs_temp_bytes(n);
            return hipSuccess;
        }
        return rs_sort_pairs(tmp, keys_in, keys_out, vals_in, vals_out, n, bits, stream);
    }
    if (n > radix_min)
        return rocprim::radix_sort_pairs<radix_only>(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n,
                                                     0u, bits, stream);
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, bits, stream);
}

}  // namespace wm

#endif  // WM_SORT_HPP
