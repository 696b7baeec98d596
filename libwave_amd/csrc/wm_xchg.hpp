// wm_xchg.hpp -- the sharded loop's exchange step INSIDE the kernel that needs its result: every rank's block of
// kBlkLen doubles written straight into every peer's mailbox over xGMI, and added up, in rank order, by the one
// workgroup that goes on to solve.  No launch of its own, no collective kernel between a rank's sums and its solve.
//
// The reference has nothing here (its only parallelism is one matcher per thread, multi_matcher.hpp:32); north_star
// asks for "RCCL all-reduce over xGMI of the normal equations only".  ncclAllReduce stays the exchange wherever
// mailboxes cannot be set up (wm_shard.hip: comm_allreduce); where they can, 272 bytes per rank do not need a
// collective kernel's launch and its ring protocol: xGMI is point to point, a rank's 68 stores reach each peer in one hop.
//
// Protocol (the low-latency one RCCL's own LL kernels use): a block travels as 2 * kBlkLen eight-byte words
// {round : 32 | half of a double : 32}.  Eight-byte stores are single-copy atomic, so a word is either the one of
// this round or an older one -- the reader polls each word until its tag is the round's and needs no fence and no
// flag that would have to be ordered behind the data.  Two buffers (round parity): a rank can be at most one round
// ahead of the slowest one, because finishing round k needs every rank's round-k block.  The round counter is a
// word in each rank's own device memory, advanced by the kernel itself: ranks execute the same sequence of exchanges
// (iterations behind a `done` skip it on every rank alike -- the state they test is the same solve's on every rank).
//
// Mailbox memory is fine-grained / uncached device memory (hipExtMallocWithFlags): remote stores must be visible to a
// kernel that is already running, and system-scope loads of it must not be served from this device's L2.
#pragma once
#include "wm_internal.hpp"

namespace wm {

constexpr int kXWords = 2 * kBlkLen;  // words of one block
constexpr int kXMaxWorld = 16;        // ranks whose halves fit the solve kernel's LDS (one node: 8)

// a rank's mailbox: [2 parities][world][kXWords] words, then the round counter (one word)
__host__ __device__ inline size_t xchg_mailbox_words(int world) { return (size_t) 2 * (size_t) world * kXWords + 8; }

struct XchgDev {
    unsigned long long *const *peer;  // [world] (device memory): rank r's mailbox as THIS device addresses it
    unsigned long long *mine;         // this rank's own mailbox (= peer[rank])
    int world, rank;
    unsigned timeout_ms;              // a peer that never delivers: give up (the registration fails, it does not hang)
};

#if defined(__HIPCC__)
// One all-reduce (sum, rank order) of the kBlkLen doubles in s_in (LDS) into s_out (LDS), by ALL threads of ONE
// workgroup; s_half: world * kXWords words of LDS.  Returns false (uniformly) if a peer's block did not arrive in time.
template <int THREADS>
__device__ __forceinline__ bool xchg_allreduce(const XchgDev &x, const double *s_in, double *s_out, unsigned *s_half,
                                               unsigned *s_ctl /* [2] LDS: round, failed */) {
    if (threadIdx.x == 0) {
        unsigned long long *counter = x.mine + (size_t) 2 * (size_t) x.world * kXWords;
        const unsigned long long r = *counter + 1ull;
        *counter = r;
        s_ctl[0] = (unsigned) r == 0u ? 1u : (unsigned) r;  // (tag 0 is "never written")
        s_ctl[1] = 0u;
    }
    __syncthreads();
    const unsigned round = s_ctl[0];
    const size_t slot = (size_t) (round & 1u) * (size_t) x.world;
    const int n = x.world * kXWords;
    for (int t = threadIdx.x; t < n; t += THREADS) {  // this rank's block into every mailbox (its own too)
        const int r = t / kXWords, w = t - r * kXWords;
        const unsigned long long bits = (unsigned long long) __double_as_longlong(s_in[w >> 1]);
        const unsigned half = (w & 1) ? (unsigned) (bits >> 32) : (unsigned) bits;
        unsigned long long *dst = x.peer[r] + (slot + (size_t) x.rank) * kXWords + w;
        __hip_atomic_store(dst, ((unsigned long long) round << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const unsigned long long t0 = wall_clock64();  // (100 MHz)
    const unsigned long long limit = (unsigned long long) x.timeout_ms * 100000ull;
    for (int t = threadIdx.x; t < n; t += THREADS) {  // every rank's block out of this rank's mailbox
        const int r = t / kXWords, w = t - r * kXWords;
        const unsigned long long *src = x.mine + (slot + (size_t) r) * kXWords + w;
        unsigned long long v;
        for (unsigned spin = 0;; ++spin) {
            v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned) (v >> 32) == round) break;
            // (a time limit of zero -- tests -- gives up on the first look that finds nothing)
            if ((limit == 0ull || (spin & 63u) == 63u) && wall_clock64() - t0 > limit) {
                s_ctl[1] = 1u;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        s_half[t] = (unsigned) v;
    }
    __syncthreads();
    if (threadIdx.x < kBlkLen) {
        const int c = threadIdx.x;
        double s = 0.0;
        for (int r = 0; r < x.world; ++r) {  // rank order: the same bits on every rank
            const unsigned lo = s_half[r * kXWords + 2 * c], hi = s_half[r * kXWords + 2 * c + 1];
            const double v = __longlong_as_double((long long) (((unsigned long long) hi << 32) | lo));
            s = r == 0 ? v : s + v;
        }
        s_out[c] = s;
    }
    __syncthreads();
    return s_ctl[1] == 0u;
}
#endif

}  // namespace wm
