// wm_solve_tail.hpp -- the iteration's solve WITHOUT a launch of its own (round 6).
//
// An ICP iteration used to end in two or three dependent launches: the search kernel (one row of partial sums per
// workgroup), k_reduce_rows (128 rows -> 1, full searches only) and k_reduce_solve (one workgroup: rows -> sums ->
// pcl::umeyama -> DefaultConvergenceCriteria -> the iteration's record).  Each dependent launch is ~4.5 us of
// dispatch before its first instruction; 50 iterations paid 75 of them (0.57 ms of a 3.75 ms registration).
// Here the LAST workgroup to finish does that work in place -- the recipe k_ndt_derivs proved in round 5:
//   * a workgroup's row is stored at agent scope (written through to where every XCD sees it),
//   * the storing wave waits for its own stores (s_waitcnt vmcnt(0)), then draws a ticket with a RELAXED
//     agent-scope atomic: no fence (an agent-scope release writes an XCD's whole L2 back; with __threadfence()
//     around the tickets an NDT pass took 157 us instead of 89),
//   * whoever draws the last ticket reads the rows at agent scope (past its own XCD's L2) and goes on.
// Groups and orders are fixed by the workgroup NUMBERS, never by who finishes when: the sums are bit-reproducible.
// Two levels for kernels of many workgroups (the last of every kTailGroup consecutive workgroups adds that group's
// rows into one; the last of those adds the groups' rows and solves): one workgroup adding a thousand rows is a
// chain of dependent round trips to the memory-side cache.
//
// What replaces: k_reduce_solve<3> behind k_reduce_rows (full searches) and behind k_nn_cert / k_nn_list (certified
// iterations) of the unsharded loop; the sharded loop keeps its own kernels (its exchange sits between sums and solve).
// (pcl::IterativeClosestPoint::computeTransformation's per-iteration tail, wave_matching/src/icp.cpp:95,116,126.)
#ifndef WM_SOLVE_TAIL_HPP
#define WM_SOLVE_TAIL_HPP

#include "wm_internal.hpp"
#include "wm_math.hpp"
#include "wm_icp_step.hpp"

namespace wm {

constexpr unsigned kTailGroup = 32;  // workgroups per first-level group of a two-level tail
constexpr int kTailRow = 20;         // doubles per row of a tail: the kAcc sums, [kAcc] = queries this workgroup searched, one spare
constexpr unsigned kTailTickets = 1024;  // ticket words: [0] the final level, [1 + g] group g (up to 1023 groups = 32 736 workgroups)

struct TailArgs {
    unsigned *ticket;         // kTailTickets words, all zero between launches (the last finishers put the zeros back)
    double *grows;            // the groups' rows (two-level), kTailRow doubles each
    unsigned long long *pub;  // pinned: the iterations' records (publish_step)
    int pub_slots;
};

// What the host steers by while it runs ahead of the device (icp_run_loop): one 8-byte word in pinned memory -- done
// flag, iterations finished, the step's size -- in ONE system-scope store (pub[0]: the latest; pub[k]: iteration k's
// own record, so that what the host decides from does not depend on when it looks).
__device__ __forceinline__ void publish_step(const IcpDevState *s, unsigned long long *pub, int pub_slots) {
    if (!pub) return;
    // [iteration : 16 | step size as bfloat16 : 16 | changed matches : 16 | searched by the certificate kernel : 16]
    // -- fractions in 1 / 65535
    const unsigned f_ch = (unsigned) (fminf(fmaxf(s->frac_changed, 0.f), 1.f) * 65535.f + 0.5f);
    const unsigned f_un = (unsigned) (fminf(fmaxf(s->frac_unsettled, 0.f), 1.f) * 65535.f + 0.5f);
    const unsigned long long w = ((unsigned long long) ((unsigned) s->iter & 0xFFFFu) << 48) |
                                 ((unsigned long long) (__float_as_uint(s->step_disp) >> 16) << 32) |
                                 ((unsigned long long) f_ch << 16) | (unsigned long long) f_un;
    if (s->iter >= 1 && s->iter <= pub_slots) __hip_atomic_store(pub + s->iter, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // ([0]: bit 0 = done, above it the number of iterations finished by then -- ONE word, so that a host that sees
    // `done` before the last record knows whether that record is still to come)
    __hip_atomic_store(pub, s->done ? (1ull | ((unsigned long long) (unsigned) s->iter << 1)) : 0ull, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int THREADS>
struct TailLds {
    static constexpr int kLanes = THREADS / kTailRow;  // row-lanes of kTailRow columns
    IcpDevState st;
    double part[kLanes][kTailRow];
    double tot[kTailRow];
    unsigned last;
};

__device__ __forceinline__ double ld_agent_f64(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent_f64(double *p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Column sums of `n` rows (kTailRow doubles apart, read at agent scope), all THREADS threads: thread (g, c) adds rows
// g, g + kLanes, ... of column c, eight loads in flight; then thread c adds the kLanes partial sums in order.
// Afterwards S.tot[c] holds column c (a barrier has been passed).
template <int THREADS>
__device__ __forceinline__ void tail_add_rows(const double *rows, unsigned n, TailLds<THREADS> &S) {
    constexpr unsigned kLanes = (unsigned) TailLds<THREADS>::kLanes;
    const unsigned t = threadIdx.x, c = t % (unsigned) kTailRow, g = t / (unsigned) kTailRow;
    if (g < kLanes) {
        double a = 0.0;
        constexpr unsigned U = 8;
        for (unsigned r0 = g; r0 < n; r0 += U * kLanes) {
            double v[U];
#pragma unroll
            for (unsigned u = 0; u < U; ++u) {
                const unsigned r = r0 + u * kLanes;
                v[u] = r < n ? ld_agent_f64(rows + (size_t) r * kTailRow + c) : 0.0;
            }
#pragma unroll
            for (unsigned u = 0; u < U; ++u) a += v[u];
        }
        S.part[g][c] = a;
    }
    __syncthreads();
    if (t < (unsigned) kTailRow) {
        double r = 0.0;
#pragma unroll
        for (unsigned gg = 0; gg < kLanes; ++gg) r += S.part[gg][t];
        S.tot[t] = r;
    }
    __syncthreads();
}

// The solve from the summed row in S.tot, by the workgroup that got there last: state staged in LDS by all threads (one
// round trip instead of two dozen dependent ones by the solving lane), icp_apply_stats in thread 0, the record
// published, the state written back whole.  The state in memory was last written by the previous iteration's tail
// (a kernel boundary ago) and is touched by nobody else while this kernel runs.
template <int THREADS>
__device__ __forceinline__ void tail_solve(IcpDevState *st, const TailArgs &ta, TailLds<THREADS> &S) {
    static_assert(sizeof(IcpDevState) % 4 == 0, "word-wise staging");
    constexpr unsigned kWords = sizeof(IcpDevState) / 4;
    const unsigned long long t_start = clock64();
    for (unsigned w = threadIdx.x; w < kWords; w += THREADS)
        reinterpret_cast<unsigned *>(&S.st)[w] = reinterpret_cast<const unsigned *>(st)[w];
    __syncthreads();
    if (threadIdx.x == 0) {
        S.st.dbg[0] = t_start;
        S.st.dbg[1] = clock64();
        double a[kAcc], ex[kStatsLen];
#pragma unroll
        for (int k = 0; k < kAcc; ++k) a[k] = S.tot[k];
        expand_stats(S.st.mode, a, ex, S.st.changed_mask);
        S.st.local_handled = ex[kStatsLen - 1];
#pragma unroll
        for (int k = 0; k < kStatsLen; ++k) S.st.stats[k] = ex[k];
        S.st.dbg[2] = clock64();
        icp_apply_stats(&S.st, ex, (long long) S.tot[kAcc]);
        publish_step(&S.st, ta.pub, ta.pub_slots);
        S.st.dbg[3] = clock64();
    }
    __syncthreads();
    for (unsigned w = threadIdx.x; w < kWords; w += THREADS)
        reinterpret_cast<unsigned *>(st)[w] = reinterpret_cast<const unsigned *>(&S.st)[w];
}

// To be called by ALL threads of EVERY workgroup of the kernel, after threads of WAVE 0 have stored the workgroup's row
// `my_row` of `rows` (kTailRow doubles, st_agent_f64).  Returns in all but one workgroup right after the ticket.
// two_level: rows are grouped kTailGroup at a time.
template <int THREADS>
__device__ __forceinline__ void tail_finish(double *rows, unsigned my_row, unsigned nrows, bool two_level, IcpDevState *st,
                                            const TailArgs &ta, TailLds<THREADS> &S) {
    const unsigned t = threadIdx.x;
    if (two_level) {
        const unsigned grp = my_row / kTailGroup, ngroups = (nrows + kTailGroup - 1u) / kTailGroup;
        const unsigned members = min(kTailGroup, nrows - grp * kTailGroup);
        if (t < 64u) {  // (the row's writers are lanes of wave 0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (t == 0)
                S.last = __hip_atomic_fetch_add(ta.ticket + 1u + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1u ? 1u : 0u;
        }
        __syncthreads();
        if (!S.last) return;
        tail_add_rows<THREADS>(rows + (size_t) grp * kTailGroup * kTailRow, members, S);
        if (t < (unsigned) kTailRow) st_agent_f64(ta.grows + (size_t) grp * kTailRow + t, S.tot[t]);
        if (t < 64u) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (t == 0) {
                __hip_atomic_store(ta.ticket + 1u + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (for the next launch)
                S.last = __hip_atomic_fetch_add(ta.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1u ? 1u : 0u;
            }
        }
        __syncthreads();
        if (!S.last) return;
        tail_add_rows<THREADS>(ta.grows, ngroups, S);
    } else {
        if (t < 64u) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (t == 0)
                S.last = __hip_atomic_fetch_add(ta.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nrows - 1u ? 1u : 0u;
        }
        __syncthreads();
        if (!S.last) return;
        tail_add_rows<THREADS>(rows, nrows, S);
    }
    if (t == 0) __hip_atomic_store(ta.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tail_solve<THREADS>(st, ta, S);
}

}  // namespace wm

#endif  // WM_SOLVE_TAIL_HPP
