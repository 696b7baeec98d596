// wm_nn.hip -- exact 1-nearest-neighbour correspondence search on gfx950.
//
// Replaces pcl::registration::CorrespondenceEstimation::determineCorrespondences
// (one FLANN kd-tree 1-NN query per source point per ICP iteration), which is
// where ~95 % of the reference's ICPMatcher::match() time goes
// (wave_matching/src/icp.cpp:95,116,126 -> icp.align()).
//
// Contract (what a FLANN exact search + PCL's distance gate give): for every
// source point the target point with the smallest float squared distance
//     d2 = (dx*dx + dy*dy) + dz*dz            (no FMA contraction)
// provided d2 <= max_corr^2; ties resolve to the lowest target index.  A result
// is carried as one 64-bit key (d2 bits << 32 | target index): non-negative
// floats order like unsigned ints, so "min over keys" is the exact arg-min with
// the tie rule built in, and "no match" is the initial key (threshold, ~0).
//
// Two kernels:
//   k_nn_grid   one lane per query: a certified radius search over a ladder of uniform
//               grids (cell size x2 per level); small radii are scanned by the query's own
//               lane, large ones by the whole wavefront (see the kernel's comment).
//   k_nn_brute  LDS-tiled all-pairs search (small clouds / cross-check).
#include "wm_internal.hpp"
#include "wm_icp_step.hpp"
#include "wm_bins.hpp"
#include "wm_wave.hpp"

#include <atomic>

namespace wm {

__device__ __forceinline__ unsigned long long make_key(float d2, unsigned idx) {
    return ((unsigned long long) __float_as_uint(d2) << 32) | idx;
}

__device__ __forceinline__ float canon_d2(float qx, float qy, float qz, const float4 &t) {
    const float dx = qx - t.x, dy = qy - t.y, dz = qz - t.z;
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// PCL's float transform of a source point: ((m00*x + m01*y) + m02*z) + m03
__device__ __forceinline__ void xform(const float *T, const float4 &p, float &x, float &y,
                                      float &z) {
    x = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[0], p.x), __fmul_rn(T[1], p.y)),
                            __fmul_rn(T[2], p.z)), T[3]);
    y = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[4], p.x), __fmul_rn(T[5], p.y)),
                            __fmul_rn(T[6], p.z)), T[7]);
    z = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[8], p.x), __fmul_rn(T[9], p.y)),
                            __fmul_rn(T[10], p.z)), T[11]);
}

// Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8).
// Remap so each XCD works on one contiguous (Morton-compact) eighth of the
// queries and its private L2 holds one spatial region of the target.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nblocks) {
    const unsigned per = (nblocks + 7u) / 8u;
    return (b & 7u) * per + (b >> 3);
}
// The one used: the XCDs take turns in chunks of S blocks (nblocks a multiple of 8 S) of the
// Morton-ordered queries.  A chunk of 32 blocks = 2048 queries is still one compact region for the
// XCD's L2, but a region of EXPENSIVE queries (the far corner of a rotated cloud in the early
// iterations) is now shared by all eight XCDs instead of landing on the one that owns that eighth
// of the cloud: 83.0 -> 79.4 us per launch on the 1M pair (chunks of 8-32 equal, 128: 80.0,
// 512: 84.5, whole eighths: 83.0).
__device__ __forceinline__ unsigned xcd_remap_chunked(unsigned b, unsigned S) {
    const unsigned x = b & 7u, l = b >> 3;
    const unsigned chunk = l / S;
    return (chunk * 8u + x) * S + (l - chunk * S);
}

// The grid tables are reached through pointers read from memory, so the compiler only knows
// them as generic (flat) addresses; they always point into HBM -- say so, and get global_load
// instead of flat_load (no LDS-aperture check, no lgkmcnt coupling).
typedef float f4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f4v *gp_f4;
typedef const __attribute__((address_space(1))) unsigned *gp_u32;
__device__ __forceinline__ float4 ldp(const float4 *p, size_t j) {
    const f4v v = ((gp_f4) p)[j];
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ unsigned ldc(const unsigned *p, size_t j) { return ((gp_u32) p)[j]; }
// Result stores.  `nt`: non-temporal -- the line does not stay (dirty) in the XCD's L2, so the kernel
// boundary behind the search has no write-back to wait for (a search leaves 12-26 MB of results that
// nothing on this XCD reads again before the next iteration)
__device__ __forceinline__ void st_f4(float4 *p, float x, float y, float z, float w, bool nt) {
    f4v v = {x, y, z, w};
    if (nt) __builtin_nontemporal_store(v, (f4v *) p);
    else *(f4v *) p = v;
}
__device__ __forceinline__ void st_u64(unsigned long long *p, unsigned long long v, bool nt) {
    if (nt) __builtin_nontemporal_store(v, p);
    else *p = v;
}
__device__ __forceinline__ void st_f32(float *p, float v, bool nt) {
    if (nt) __builtin_nontemporal_store(v, p);
    else *p = v;
}
__device__ __forceinline__ void st_f64(double *p, double v, bool nt) {
    if (nt) __builtin_nontemporal_store(v, p);
    else *p = v;
}
__device__ __forceinline__ float canon_d2v(float qx, float qy, float qz, const f4v &t) {
    return canon_d2(qx, qy, qz, make_float4(t.x, t.y, t.z, t.w));
}

// ------------------------------------------------------------- grid search
// scan the contiguous run [s, e) of cell-sorted target points, four loads in flight.
// The last group may read up to three entries past e: they are the next cells' points (real
// target points -- a closer one among them is a legitimate find) or the NaN padding at the end
// of the array (a NaN distance has the largest key and never wins), so no clamping is needed
// and the four loads share one address.
__device__ __forceinline__ unsigned long long scan_run(const float4 *__restrict__ pts, unsigned s,
                                                       unsigned e, float qx, float qy, float qz,
                                                       unsigned long long best) {
    for (unsigned j = s; j < e; j += 4) {
        const gp_f4 p = (gp_f4) pts + j;
        const f4v t0 = p[0], t1 = p[1], t2 = p[2], t3 = p[3];
        const unsigned long long k0 = make_key(canon_d2v(qx, qy, qz, t0), __float_as_uint(t0.w));
        const unsigned long long k1 = make_key(canon_d2v(qx, qy, qz, t1), __float_as_uint(t1.w));
        const unsigned long long k2 = make_key(canon_d2v(qx, qy, qz, t2), __float_as_uint(t2.w));
        const unsigned long long k3 = make_key(canon_d2v(qx, qy, qz, t3), __float_as_uint(t3.w));
        const unsigned long long a = k0 < k1 ? k0 : k1, b = k2 < k3 ? k2 : k3;
        const unsigned long long m = a < b ? a : b;
        best = m < best ? m : best;
    }
    return best;
}

// Scan the target points that can lie inside ball(q, min(r, sqrt(best))) on level g and report
// the distance `margin` from the query to the faces of the box of cells covering
// [q - r, q + r]^3: every point NOT scanned is either farther than `margin` or farther than
// the best distance at the time it was skipped.
//   * the box is walked row by row ((y,z) rows; cells are x-fastest, so a row is one
//     contiguous run of the cell-sorted array), kRowChunk rows at a time: the chunk's
//     cell_start look-ups are issued together (one memory round trip per chunk, not per row);
//   * a row is cut down to the chord of ball(q, sqrt(best)) -- with rho the (y,z) distance of
//     the row, a point of the row closer than Rb has
//         |x - qx| <= sqrt(Rb^2 - (rho - slack)^2) <= sqrt(Rb^2 - rho^2 + 2 slack (Rb + slack))
//     (cell units; Rb inflated by 1e-5 against the approximate hardware square roots): rows
//     outside the ball cost nothing, rows near its rim a cell or two.
// Pruning changes the work, never the result.
constexpr int kRowChunk = 6;
constexpr int kLayeredRows = 18;  // boxes with more (y,z) rows than this are walked layer by layer
template <bool COST>
__device__ __forceinline__ unsigned long long scan_box(const GridDev &g, float qx, float qy,
                                                       float qz, float r, unsigned long long best,
                                                       float *margin, uint2 *runs, unsigned lane,
                                                       bool allow_layered, unsigned &cost) {
    const float big = 4.0e6f;  // clamp in float so far-away queries cannot overflow the int cast
    const float fx = fminf(fmaxf((qx - g.ox) * g.inv_h, -big), big);
    const float fy = fminf(fmaxf((qy - g.oy) * g.inv_h, -big), big);
    const float fz = fminf(fmaxf((qz - g.oz) * g.inv_h, -big), big);
    const float rc = r * g.inv_h + g.slack;
    const int x0 = (int) floorf(fx - rc), x1 = (int) floorf(fx + rc);
    const int y0 = (int) floorf(fy - rc), y1 = (int) floorf(fy + rc);
    const int z0 = (int) floorf(fz - rc), z1 = (int) floorf(fz + rc);
    const float mx = fminf(fx - (float) x0, (float) (x1 + 1) - fx);
    const float my = fminf(fy - (float) y0, (float) (y1 + 1) - fy);
    const float mz = fminf(fz - (float) z0, (float) (z1 + 1) - fz);
    *margin = (fminf(mx, fminf(my, mz)) - g.slack) * g.h;
    const int xa = max(x0, 0), xb = min(x1, g.nx - 1);
    const int ya = max(y0, 0), yb = min(y1, g.ny - 1);
    const int za = max(z0, 0), zb = min(z1, g.nz - 1);
    if (xa > xb || ya > yb || za > zb) return best;
    // look-ups of one chunk of rows (addresses a0/a1), then the walk over its non-empty runs
    auto lookup_and_walk = [&](const unsigned (&a0)[kRowChunk], const unsigned (&a1)[kRowChunk]) {
        unsigned rs[kRowChunk], re[kRowChunk];
#pragma unroll
        for (int u = 0; u < kRowChunk; ++u) {
            rs[u] = ldc(g.cell_start, a0[u]);
            re[u] = ldc(g.cell_start, a1[u]);
        }
        // The chunk's non-empty runs go into this lane's column of an LDS list, and the lane
        // walks its own list: it moves on to its next run as soon as the current one is done,
        // so the wave makes max-over-lanes(sum of a lane's trips) trips, not
        // sum-over-runs(max-over-lanes).  With sparse rows (far queries: most rows of the ball
        // are empty) that is several times fewer.  Four points per trip, one address (reads
        // past a run's end are harmless, see scan_run).  No barrier: a lane only reads back
        // what it wrote itself, and LDS operations of one wave execute in order.
        unsigned cnt = 0;
        if constexpr (COST) cost += 1u << 16;  // (developer statistics: chunks in bits 16-23, trips below)
#pragma unroll
        for (int u = 0; u < kRowChunk; ++u)
            if (re[u] > rs[u]) runs[cnt++ * 64u + lane] = make_uint2(rs[u], re[u]);
        // ONE flat loop (a nested per-run loop would make the lanes wait for each other at
        // every run boundary again)
        unsigned idx = 0, j = 0, e = 0;
        if (cnt) {
            const uint2 run = runs[lane];
            j = run.x;
            e = run.y;
        }
        while (j < e) {
            const gp_f4 p = (gp_f4) g.pts + j;
            const f4v t0 = p[0], t1 = p[1], t2 = p[2], t3 = p[3];
            const unsigned long long k0 = make_key(canon_d2v(qx, qy, qz, t0), __float_as_uint(t0.w));
            const unsigned long long k1 = make_key(canon_d2v(qx, qy, qz, t1), __float_as_uint(t1.w));
            const unsigned long long k2 = make_key(canon_d2v(qx, qy, qz, t2), __float_as_uint(t2.w));
            const unsigned long long k3 = make_key(canon_d2v(qx, qy, qz, t3), __float_as_uint(t3.w));
            const unsigned long long a = k0 < k1 ? k0 : k1, b = k2 < k3 ? k2 : k3;
            const unsigned long long m = a < b ? a : b;
            best = m < best ? m : best;
            j += 4;
            if constexpr (COST) cost += 1u;
            if (j >= e && ++idx < cnt) {
                const uint2 run = runs[idx * 64u + lane];
                j = run.x;
                e = run.y;
            }
        }
    };
    // Big boxes (queries still far from their neighbour: dozens of rows, most of them empty
    // space) are walked layer by layer, the z-layers in lock-step across the wave: what depends
    // on the layer only (its z distance, the y chord of the ball in it, its base address) is
    // computed once per layer, and only the rows inside the y chord are enumerated at all.
    const bool layered =
        allow_layered && __popcll(__ballot((yb - ya + 1) * (zb - za + 1) > kLayeredRows)) >= 8;
    if (layered) {
        for (int kz = 0;; ++kz) {
            const int zz = za + kz;
            const bool zact = zz <= zb;
            if (__ballot(zact) == 0ull) break;
            const float Rb =
                __builtin_amdgcn_sqrtf(__uint_as_float((unsigned) (best >> 32))) * g.inv_h * 1.00001f;
            const float lim = Rb + g.slack;
            const float lim2 = lim * lim, c0 = Rb * Rb + 2.f * g.slack * lim;
            const float rz = fmaxf(fmaxf((float) zz - fz, fz - (float) (zz + 1)), 0.f);
            const float rz2 = rz * rz;
            // rows of this layer that can touch the ball: their y distance is <= sqrt(lim^2 - rz^2)
            const float hy = __builtin_amdgcn_sqrtf(fmaxf(lim2 - rz2, 0.f)) * 1.00001f;
            const bool zin = zact && !(rz2 > lim2);
            const int yl = zin ? max(ya, __float2int_rd(fy - hy)) : 1;
            const int yh = zin ? min(yb, __float2int_rd(fy + hy)) : 0;
            const unsigned basez = (unsigned) zz * g.ny * g.nx;
            for (int y0 = yl; __ballot(y0 <= yh) != 0ull; y0 += kRowChunk) {
                unsigned a0[kRowChunk], a1[kRowChunk];
#pragma unroll
                for (int u = 0; u < kRowChunk; ++u) {
                    const int yy = y0 + u;
                    const float yf = (float) yy;
                    const float ry = fmaxf(fmaxf(yf - fy, fy - (yf + 1.f)), 0.f);
                    const float rho2 = ry * ry + rz2;
                    const float hx = __builtin_amdgcn_sqrtf(fmaxf(c0 - rho2, 0.f)) * 1.00001f + g.slack;
                    const int xl = max(xa, __float2int_rd(fx - hx)), xh = min(xb, __float2int_rd(fx + hx));
                    const bool ok = yy <= yh && !(rho2 > lim2) && xl <= xh;
                    const unsigned base = basez + (unsigned) yy * g.nx;
                    a0[u] = ok ? base + xl : 0u;
                    a1[u] = ok ? base + xh + 1 : 0u;
                }
                lookup_and_walk(a0, a1);
            }
        }
        return best;
    }
    int yy = ya, zz = za;  // row cursor
    while (zz <= zb) {
        const float Rb =
            __builtin_amdgcn_sqrtf(__uint_as_float((unsigned) (best >> 32))) * g.inv_h * 1.00001f;
        const float lim = Rb + g.slack;
        const float lim2 = lim * lim, c0 = Rb * Rb + 2.f * g.slack * lim;
        // addresses first, then all look-ups back to back and unconditional (a row outside the
        // ball reads cell_start[0] twice: an empty run) -- with predicated loads the compiler
        // interleaves address arithmetic, branches and waits, and the twelve look-ups of a
        // chunk no longer overlap
        unsigned a0[kRowChunk], a1[kRowChunk];
#pragma unroll
        for (int u = 0; u < kRowChunk; ++u) {
            // distance from the query to row (yy, zz) along y and z, in cells: positive on the
            // far side, 0 inside the query's own row (branch-free form of the three cases)
            const float ry = fmaxf(fmaxf((float) yy - fy, fy - (float) (yy + 1)), 0.f);
            const float rz = fmaxf(fmaxf((float) zz - fz, fz - (float) (zz + 1)), 0.f);
            const float rho2 = ry * ry + rz * rz;
            const float hx = __builtin_amdgcn_sqrtf(fmaxf(c0 - rho2, 0.f)) * 1.00001f + g.slack;
            const int xl = max(xa, __float2int_rd(fx - hx)), xh = min(xb, __float2int_rd(fx + hx));
            const bool ok = zz <= zb && !(rho2 > lim2) && xl <= xh;
            const unsigned base = ((unsigned) zz * g.ny + yy) * g.nx;
            a0[u] = ok ? base + xl : 0u;
            a1[u] = ok ? base + xh + 1 : 0u;
            const bool wrap = yy >= yb;
            yy = wrap ? ya : yy + 1;
            zz += wrap;
        }
        lookup_and_walk(a0, a1);
    }
    return best;
}

// scan_run with runner-up tracking: `second` = d2 bits of the closest point seen other than the best
// (meeting the best again -- the seed, a point read past a run's end -- changes nothing)
__device__ __forceinline__ unsigned long long scan_run_bound(const float4 *__restrict__ pts, unsigned s, unsigned e,
                                                             float qx, float qy, float qz, unsigned long long best,
                                                             unsigned &second) {
    for (unsigned j = s; j < e; j += 4) {
        const gp_f4 p = (gp_f4) pts + j;
        const f4v t[4] = {p[0], p[1], p[2], p[3]};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long k = make_key(canon_d2v(qx, qy, qz, t[u]), __float_as_uint(t[u].w));
            if (k < best) {
                second = min(second, (unsigned) (best >> 32));
                best = k;
            } else if (k != best) {
                second = min(second, (unsigned) (k >> 32));
            }
        }
    }
    return best;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off);
        v = o < v ? o : v;
    }
    return v;
}

__device__ __forceinline__ float rl_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ unsigned rl_u(unsigned v, int lane) {
    return (unsigned) __builtin_amdgcn_readlane((int) v, lane);
}

// ------------------------------------------------- balanced walk (wave-level work sharing)
// The lane scan above makes a wavefront wait for its slowest lane: in the aligned state a query
// needs 3.3 trips of the candidate loop on average but the slowest of 64 needs 8.4 (measured:
// scripts/dev/dev_cost_model.py), so 60 % of the lanes idle through the loop that is most of the
// kernel -- on the vector ALU and on the L1 address path alike.  Here the wavefront pools the work
// instead: every lane lists its trips (four consecutive points of one of its runs) in LDS, and
// all 64 lanes then take trips off the pooled list, whoever's they are -- ceil(sum / 64) rounds
// instead of max-over-lanes.  A trip's result goes to its query's slot by an LDS atomic min on
// the 64-bit key, so the order in which candidates are seen still does not matter: same results.
// This needs all control flow around the walk to be wave-uniform (a lane that has finished its
// own search keeps working on the others'): the pass and row loops run while ANY lane has work,
// and a lane without work contributes empty rows.
constexpr int kBalCap = 1024;  // pooled trips per chunk of rows; beyond that (rare) every lane walks its own
// rows per chunk of the balanced walk: 2 / 3 / 4 / 6 / 8 / 12 rows measured 73.1 / 70.8 / 72.7 / 73.5 /
// 79.5 / 101 us per launch on the 1M pair (more rows per chunk = more registers and, with the walk
// balanced anyway, nothing gained from batching more look-ups)
constexpr int kBalRowChunk = 3;
struct BalLds {                // per wavefront
    unsigned items[kBalCap + 1];    // (owner lane << 26) | offset of the trip's first point; [kBalCap] = dump slot
    float4 q[64];                   // the queries
    unsigned long long base[64];    // address of the point array (level) each query scans
    unsigned long long best[64];    // running arg-min per query
    // values a lane needs again only after its search (the key it started from, its seed's
    // coordinates): parked here instead of in five registers the compiler would spill to scratch
    unsigned long long seeded[64];
    float bq[3][64];
};

// inclusive prefix sum over the 64 lanes (DPP row shifts inside rows of 16, then the row totals
// are broadcast into the following rows: no LDS traffic, six dependent VALU steps)
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v) {
    v += (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, true);  // row_shr:8
    v += (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}

// ubase: the point array all live lanes scan when they are on the same level (the usual case),
// nullptr when the levels differ (then L.base[owner] says which)
template <bool B>
struct BalHolder {
    BalLds v;
};
template <>
struct BalHolder<false> {
    int unused;
};

// Runner-up tracking (BOUND): besides the arg-min the search then also reports a lower bound s of
// the distance from the query to every target point OTHER than its match -- the smaller of the
// second-smallest distance it saw and the radius it pruned with.  While a later pose moves the
// query by less than s - |q - match| the match is still the nearest neighbour and no search is
// needed (k_nn_cert).  To make s useful the scan prunes with min(runner-up, best + pad) instead of
// best: `pad` is how much room the caller wants (a few times the query's last displacement).
struct Bound {
    unsigned second;  // d2 bits of the runner-up so far (a point other than the current best)
    float pad;        // metres
    unsigned *lds;    // [64] words of LDS: the runner-ups while the wave's pooled walk is under way
    float4 *win;      // [64] LDS slots: coordinates (+ index bits) of a candidate that became a query's best
    bool ok;          // false once a path without runner-up tracking has been taken: no bound to offer
    // squared prune radius, as float bits
    __device__ __forceinline__ float prune_r(unsigned long long best) const {
        const float b = __builtin_amdgcn_sqrtf(__uint_as_float((unsigned) (best >> 32))) * 1.00001f + pad;
        return fminf(b, __builtin_amdgcn_sqrtf(__uint_as_float(second)) * 1.00001f);
    }
};

// The rounds of a pooled walk: all 64 lanes take trips (owner lane << 26 | offset of four consecutive
// points) off `items`, whoever's they are, and merge what they find into the owner's slot.
template <bool BOUND>
__device__ __forceinline__ void pooled_rounds(BalLds &L, const unsigned *items, unsigned T, unsigned lane,
                                              const float4 *ubase, Bound *bnd) {
    for (unsigned k0 = 0; k0 < T; k0 += 64u) {
        const unsigned k = k0 + lane;
        if (k < T) {
            const unsigned it = items[k];
            const unsigned owner = it >> 26, j = it & 0x3FFFFFFu;
            const float4 q = L.q[owner];
            const gp_f4 p = (gp_f4) (ubase ? ubase : (const float4 *) L.base[owner]) + j;
            const f4v t0 = p[0], t1 = p[1], t2 = p[2], t3 = p[3];
            const float d0 = canon_d2v(q.x, q.y, q.z, t0), d1 = canon_d2v(q.x, q.y, q.z, t1);
            const float d2 = canon_d2v(q.x, q.y, q.z, t2), d3 = canon_d2v(q.x, q.y, q.z, t3);
            const unsigned m = min(min(__float_as_uint(d0), __float_as_uint(d1)), min(__float_as_uint(d2), __float_as_uint(d3)));
            if (m <= __float_as_uint(q.w)) {  // (d2 >= 0: bit order = numeric order)
                const unsigned long long k0_ = make_key(d0, __float_as_uint(t0.w)), k1_ = make_key(d1, __float_as_uint(t1.w));
                const unsigned long long k2_ = make_key(d2, __float_as_uint(t2.w)), k3_ = make_key(d3, __float_as_uint(t3.w));
                const unsigned long long a = k0_ < k1_ ? k0_ : k1_, b = k2_ < k3_ ? k2_ : k3_;
                if constexpr (BOUND) {
                    // the trip's smallest key contends for the owner's best; whichever of the two loses
                    // is a runner-up candidate, and so is the trip's own second smallest (the global
                    // runner-up is one trip's winner or the best trip's second).  Meeting the same point
                    // again (old == mine: the seed, or a point read past a run's end) changes nothing.
                    const unsigned long long mn = a < b ? a : b;
                    const unsigned u0 = __float_as_uint(d0), u1 = __float_as_uint(d1), u2 = __float_as_uint(d2),
                                   u3 = __float_as_uint(d3);
                    const unsigned lo01 = min(u0, u1), hi01 = max(u0, u1), lo23 = min(u2, u3), hi23 = max(u2, u3);
                    const unsigned sec = min(max(lo01, lo23), min(hi01, hi23));
                    const unsigned long long old = atomicMin(&L.best[owner], mn);
                    unsigned push = sec;
                    if (old != mn) push = min(push, (unsigned) ((old > mn ? old : mn) >> 32));
                    atomicMin(&bnd->lds[owner], push);
                    if (old > mn && bnd->win) {
                        // a new best: its coordinates go to the owner's slot, tagged with its index, so
                        // that the owner need not fetch them from memory afterwards (a slot written by
                        // two winners of one round may hold the loser's: the tag tells)
                        const f4v c = mn == k0_ ? t0 : (mn == k1_ ? t1 : (mn == k2_ ? t2 : t3));
                        bnd->win[owner] = make_float4(c.x, c.y, c.z, c.w);
                    }
                } else {
                    atomicMin(&L.best[owner], a < b ? a : b);
                }
            }
        }
    }
}

template <bool COST, int RC, bool BOUND = false>
__device__ __forceinline__ void balanced_walk(BalLds &L, const unsigned (&rs)[RC],
                                              const unsigned (&re)[RC], unsigned lane,
                                              const float4 *pts, const float4 *ubase, float qx, float qy,
                                              float qz, unsigned long long &best, unsigned &cost,
                                              unsigned long long *prof, bool filter, Bound *bnd = nullptr) {
    const unsigned long long prof_t0 = COST ? clock64() : 0ull;
    unsigned len[RC], t = 0, longest = 0;
#pragma unroll
    for (int u = 0; u < RC; ++u) {
        len[u] = (unsigned) max((int) (re[u] - rs[u]), 0);
        t += (len[u] + 3u) >> 2;
        longest = max(longest, len[u]);
    }
    const unsigned incl = wave_incl_scan(t);
    const unsigned T = rl_u(incl, 63);
    if (T == 0u) return;  // (wave-uniform)
    if constexpr (COST) cost += t;
    if (T > (unsigned) kBalCap) {  // too much for the list: every lane for itself
        if constexpr (BOUND) {
#pragma unroll
            for (int u = 0; u < RC; ++u) best = scan_run_bound(pts, rs[u], re[u], qx, qy, qz, best, bnd->second);
        } else {
#pragma unroll
            for (int u = 0; u < RC; ++u) best = scan_run(pts, rs[u], re[u], qx, qy, qz, best);
        }
        return;
    }
    unsigned off = incl - t;
    const unsigned tag = lane << 26;
    if (__ballot(longest > 8u) == 0ull) {
        // every run is one or two trips (the usual case): straight-line code, entries that do not
        // exist go to a dump slot past the list
#pragma unroll
        for (int u = 0; u < RC; ++u) {
            L.items[len[u] > 0u ? off : (unsigned) kBalCap] = tag | rs[u];
            L.items[len[u] > 4u ? off + 1u : (unsigned) kBalCap] = tag | (rs[u] + 4u);
            off += (len[u] + 3u) >> 2;
        }
    } else {
#pragma unroll
        for (int u = 0; u < RC; ++u)
            for (unsigned j = rs[u]; j < re[u]; j += 4u) L.items[off++] = tag | j;
    }
    L.best[lane] = best;
    // the owner's best d2 at the start of the walk rides along with its query: a worker builds the four
    // 64-bit keys and issues the LDS atomic only when one of its candidates can get under it -- rarely,
    // once the clouds are close (the seed is usually the neighbour).  A stale bound lets more through, never less.
    if constexpr (BOUND) {
        // ... under the prune radius, that is: runner-up candidates must get through too
        const float pr = bnd->prune_r(best);
        bnd->lds[lane] = bnd->second;
        reinterpret_cast<unsigned *>(&L.q[lane])[3] = __float_as_uint(pr * pr);
    } else {
        reinterpret_cast<unsigned *>(&L.q[lane])[3] = filter ? (unsigned) (best >> 32) : 0x7F800000u;
    }
    __builtin_amdgcn_wave_barrier();  // (LDS operations of one wave execute in order; this only stops the compiler)
    pooled_rounds<BOUND>(L, L.items, T, lane, ubase, bnd);
    __builtin_amdgcn_wave_barrier();
    best = L.best[lane];
    if constexpr (BOUND) bnd->second = bnd->lds[lane];
    if constexpr (COST) {
        prof[0] += clock64() - prof_t0;  // walk (list building + rounds)
        prof[1] += (T + 63u) / 64u;      // rounds
        prof[2] += 1;                    // walks
    }
}

// scan_box with wave-uniform loops (see above); `live` = this lane has a search of its own going
template <bool COST, int RC, bool BOUND = false>
__device__ __forceinline__ unsigned long long scan_box_bal(const GridDev &g, bool live, float qx, float qy,
                                                           float qz, float r, unsigned long long best,
                                                           float *margin, BalLds &L, unsigned lane,
                                                           bool allow_layered, const float4 *ubase,
                                                           unsigned &cost, unsigned long long *prof, bool filter,
                                                           Bound *bnd = nullptr) {
    const float big = 4.0e6f;
    const float fx = fminf(fmaxf((qx - g.ox) * g.inv_h, -big), big);
    const float fy = fminf(fmaxf((qy - g.oy) * g.inv_h, -big), big);
    const float fz = fminf(fmaxf((qz - g.oz) * g.inv_h, -big), big);
    const float rc = r * g.inv_h + g.slack;
    const int x0 = (int) floorf(fx - rc), x1 = (int) floorf(fx + rc);
    const int y0 = (int) floorf(fy - rc), y1 = (int) floorf(fy + rc);
    const int z0 = (int) floorf(fz - rc), z1 = (int) floorf(fz + rc);
    const float mx = fminf(fx - (float) x0, (float) (x1 + 1) - fx);
    const float my = fminf(fy - (float) y0, (float) (y1 + 1) - fy);
    const float mz = fminf(fz - (float) z0, (float) (z1 + 1) - fz);
    *margin = (fminf(mx, fminf(my, mz)) - g.slack) * g.h;
    const int xa = max(x0, 0), xb = min(x1, g.nx - 1);
    const int ya = max(y0, 0), yb = min(y1, g.ny - 1);
    const int za = max(z0, 0), zb = min(z1, g.nz - 1);
    const bool has = live && !(xa > xb || ya > yb || za > zb);
    auto lookup_and_walk = [&](const unsigned (&a0)[RC], const unsigned (&a1)[RC]) {
        unsigned rs[RC], re[RC];
#pragma unroll
        for (int u = 0; u < RC; ++u) {
            rs[u] = ldc(g.cell_start, a0[u]);
            re[u] = ldc(g.cell_start, a1[u]);
        }
        if constexpr (COST) cost += has ? 1u << 16 : 0u;
        balanced_walk<COST, RC, BOUND>(L, rs, re, lane, g.pts, ubase, qx, qy, qz, best, cost, prof, filter, bnd);
    };
    // radius (cell units) beyond which a point cannot matter: the best distance so far -- or, with
    // runner-up tracking, the prune radius
    auto ball_r = [&]() -> float {
        if constexpr (BOUND) return bnd->prune_r(best) * g.inv_h * 1.00001f;
        else return __builtin_amdgcn_sqrtf(__uint_as_float((unsigned) (best >> 32))) * g.inv_h * 1.00001f;
    };
    const bool layered =
        allow_layered && __popcll(__ballot(has && (yb - ya + 1) * (zb - za + 1) > kLayeredRows)) >= 8;
    if (layered) {
        for (int kz = 0;; ++kz) {
            const int zz = za + kz;
            const bool zact = has && zz <= zb;
            if (__ballot(zact) == 0ull) break;
            const float Rb = ball_r();
            const float lim = Rb + g.slack;
            const float lim2 = lim * lim, c0 = Rb * Rb + 2.f * g.slack * lim;
            const float rz = fmaxf(fmaxf((float) zz - fz, fz - (float) (zz + 1)), 0.f);
            const float rz2 = rz * rz;
            const float hy = __builtin_amdgcn_sqrtf(fmaxf(lim2 - rz2, 0.f)) * 1.00001f;
            const bool zin = zact && !(rz2 > lim2);
            const int yl = zin ? max(ya, __float2int_rd(fy - hy)) : 1;
            const int yh = zin ? min(yb, __float2int_rd(fy + hy)) : 0;
            const unsigned basez = (unsigned) zz * g.ny * g.nx;
            for (int yc = yl; __ballot(yc <= yh) != 0ull; yc += RC) {
                unsigned a0[RC], a1[RC];
#pragma unroll
                for (int u = 0; u < RC; ++u) {
                    const int yy = yc + u;
                    const float yf = (float) yy;
                    const float ry = fmaxf(fmaxf(yf - fy, fy - (yf + 1.f)), 0.f);
                    const float rho2 = ry * ry + rz2;
                    const float hx = __builtin_amdgcn_sqrtf(fmaxf(c0 - rho2, 0.f)) * 1.00001f + g.slack;
                    const int xl = max(xa, __float2int_rd(fx - hx)), xh = min(xb, __float2int_rd(fx + hx));
                    const bool ok = yy <= yh && !(rho2 > lim2) && xl <= xh;
                    const unsigned base = basez + (unsigned) yy * g.nx;
                    a0[u] = ok ? base + xl : 0u;
                    a1[u] = ok ? base + xh + 1 : 0u;
                }
                lookup_and_walk(a0, a1);
            }
        }
        return best;
    }
    int yy = ya, zz = has ? za : zb + 1;  // row cursor; a lane without rows is past its last one
    while (__ballot(zz <= zb) != 0ull) {
        const float Rb = ball_r();
        const float lim = Rb + g.slack;
        const float lim2 = lim * lim, c0 = Rb * Rb + 2.f * g.slack * lim;
        unsigned a0[RC], a1[RC];
#pragma unroll
        for (int u = 0; u < RC; ++u) {
            const float ry = fmaxf(fmaxf((float) yy - fy, fy - (float) (yy + 1)), 0.f);
            const float rz = fmaxf(fmaxf((float) zz - fz, fz - (float) (zz + 1)), 0.f);
            const float rho2 = ry * ry + rz * rz;
            const float hx = __builtin_amdgcn_sqrtf(fmaxf(c0 - rho2, 0.f)) * 1.00001f + g.slack;
            const int xl = max(xa, __float2int_rd(fx - hx)), xh = min(xb, __float2int_rd(fx + hx));
            const bool ok = zz <= zb && !(rho2 > lim2) && xl <= xh;
            const unsigned base = ((unsigned) zz * g.ny + yy) * g.nx;
            a0[u] = ok ? base + xl : 0u;
            a1[u] = ok ? base + xh + 1 : 0u;
            const bool wrap = yy >= yb;
            yy = wrap ? ya : yy + 1;
            zz += (wrap && zz <= zb) ? 1 : 0;
        }
        lookup_and_walk(a0, a1);
    }
    return best;
}

// The same scan for a wave with FEW searches going (the certificate kernel's late iterations: a
// handful of unsettled queries per wave, boxes of one to nine rows).  scan_box_bal makes every lane
// step through its own rows three at a time, so a wave with five live lanes still pays a full step
// -- row chords, look-ups, list, round -- per three rows of its largest box.  Here the ROWS are pooled
// too: the owners list (owner, row) pairs in LDS, every lane takes one pair -- the row's chord and its
// two cell_start look-ups, one memory round trip for the whole wave --, the runs found become the
// pooled trip list, and the rounds follow: one pass over everything, whatever the boxes' shapes.
// Runner-up tracking as in the BOUND walk; the chords are cut with the prune radius at entry.
// Returns false (nothing changed) when the job is not small: more than kRowPool rows in the wave,
// more than kRowsPerLane in one box, or more trips than the list holds; the caller then takes
// scan_box_bal.  All live lanes must be on level g.
constexpr unsigned kRowPool = 128, kRowsPerLane = 16, kRowTrips = 512;
struct RowLds {  // overlays BalLds::items (1025 words)
    unsigned trips[kRowTrips];
    unsigned map[kRowPool];   // owner lane | row number << 8
    int box[5][64];           // per owner: xa, xb, ya, za, rows per z layer
};
static_assert(sizeof(RowLds) <= sizeof(unsigned) * (kBalCap + 1), "RowLds overlays BalLds::items");

__device__ __forceinline__ bool scan_box_rows(const GridDev &g, bool live, float qx, float qy, float qz, float r,
                                              unsigned long long &best, float *margin, BalLds &L, unsigned lane,
                                              Bound *bnd) {
    const float big = 4.0e6f;
    const float fx = fminf(fmaxf((qx - g.ox) * g.inv_h, -big), big);
    const float fy = fminf(fmaxf((qy - g.oy) * g.inv_h, -big), big);
    const float fz = fminf(fmaxf((qz - g.oz) * g.inv_h, -big), big);
    const float rc = r * g.inv_h + g.slack;
    const int x0 = (int) floorf(fx - rc), x1 = (int) floorf(fx + rc);
    const int y0 = (int) floorf(fy - rc), y1 = (int) floorf(fy + rc);
    const int z0 = (int) floorf(fz - rc), z1 = (int) floorf(fz + rc);
    const int xa = max(x0, 0), xb = min(x1, g.nx - 1);
    const int ya = max(y0, 0), yb = min(y1, g.ny - 1);
    const int za = max(z0, 0), zb = min(z1, g.nz - 1);
    const bool has = live && !(xa > xb || ya > yb || za > zb);
    const int wy = yb - ya + 1;
    const unsigned nrows = has ? (unsigned) (wy * (zb - za + 1)) : 0u;
    const unsigned incl = wave_incl_scan(nrows);
    const unsigned R = rl_u(incl, 63);
    if (R > kRowPool || __ballot(nrows > kRowsPerLane) != 0ull) return false;
    {
        const float mx = fminf(fx - (float) x0, (float) (x1 + 1) - fx);
        const float my = fminf(fy - (float) y0, (float) (y1 + 1) - fy);
        const float mz = fminf(fz - (float) z0, (float) (z1 + 1) - fz);
        *margin = (fminf(mx, fminf(my, mz)) - g.slack) * g.h;
    }
    if (R == 0u) return true;
    RowLds &W = *reinterpret_cast<RowLds *>(L.items);
    const float pr = bnd->prune_r(best);
    L.best[lane] = best;
    bnd->lds[lane] = bnd->second;
    reinterpret_cast<unsigned *>(&L.q[lane])[3] = __float_as_uint(pr * pr);
    W.box[0][lane] = xa;
    W.box[1][lane] = xb;
    W.box[2][lane] = ya;
    W.box[3][lane] = za;
    W.box[4][lane] = wy;
    for (unsigned k = 0, off = incl - nrows; k < nrows; ++k) W.map[off + k] = lane | (k << 8);
    __builtin_amdgcn_wave_barrier();
    unsigned ttot = 0;  // trips listed and not yet walked (wave-uniform)
    bool fits = true;
    for (unsigned b0 = 0; b0 < R; b0 += 64u) {
        unsigned rs = 0, re = 0, owner = 0;
        if (b0 + lane < R) {
            const unsigned m = W.map[b0 + lane];
            owner = m & 63u;
            const int k = (int) (m >> 8), wyo = W.box[4][owner];
            const int yy = W.box[2][owner] + k % wyo, zz = W.box[3][owner] + k / wyo;
            const float4 q = L.q[owner];
            const float ofx = fminf(fmaxf((q.x - g.ox) * g.inv_h, -big), big);
            const float ofy = fminf(fmaxf((q.y - g.oy) * g.inv_h, -big), big);
            const float ofz = fminf(fmaxf((q.z - g.oz) * g.inv_h, -big), big);
            // the ball that matters, in cells (q.w = the owner's squared prune radius; cushions for the
            // approximate square roots as in scan_box_bal)
            const float Rb = __builtin_amdgcn_sqrtf(q.w) * g.inv_h * 1.00003f;
            const float lim = Rb + g.slack;
            const float lim2 = lim * lim, c0 = Rb * Rb + 2.f * g.slack * lim;
            const float ry = fmaxf(fmaxf((float) yy - ofy, ofy - (float) (yy + 1)), 0.f);
            const float rz = fmaxf(fmaxf((float) zz - ofz, ofz - (float) (zz + 1)), 0.f);
            const float rho2 = ry * ry + rz * rz;
            const float hx = __builtin_amdgcn_sqrtf(fmaxf(c0 - rho2, 0.f)) * 1.00001f + g.slack;
            const int xl = max(W.box[0][owner], __float2int_rd(ofx - hx)), xh = min(W.box[1][owner], __float2int_rd(ofx + hx));
            if (!(rho2 > lim2) && xl <= xh) {
                const unsigned base = ((unsigned) zz * g.ny + yy) * g.nx;
                rs = ldc(g.cell_start, base + xl);
                re = ldc(g.cell_start, base + xh + 1);
            }
        }
        const unsigned len = (unsigned) max((int) (re - rs), 0), t = (len + 3u) >> 2;
        const unsigned incl2 = wave_incl_scan(t);
        const unsigned Tb = rl_u(incl2, 63);
        if (Tb > kRowTrips) {  // (cells this crowded are not the small job this path is for)
            fits = false;
            break;
        }
        if (ttot + Tb > kRowTrips) {
            __builtin_amdgcn_wave_barrier();
            pooled_rounds<true>(L, W.trips, ttot, lane, g.pts, bnd);
            __builtin_amdgcn_wave_barrier();
            ttot = 0;
        }
        unsigned o = ttot + incl2 - t;
        for (unsigned j = rs; j < re; j += 4u) W.trips[o++] = (owner << 26) | j;
        ttot += Tb;
    }
    __builtin_amdgcn_wave_barrier();
    if (fits) pooled_rounds<true>(L, W.trips, ttot, lane, g.pts, bnd);
    __builtin_amdgcn_wave_barrier();
    // (also after a bail-out: whatever the rounds walked so far has been merged, and stays valid)
    best = L.best[lane];
    bnd->second = bnd->lds[lane];
    return fits;
}

// Wave-cooperative version of scan_box for ONE query (q, r, best are wave-uniform):
// the (y,z) rows of the box are first resolved to point ranges by up to 64 lanes in
// parallel (one memory round trip), then every row is streamed by all 64 lanes with
// coalesced float4 loads.  Used for queries far from their neighbour, whose scans
// would otherwise serialise thousands of dependent loads in one lane.
// second_out != nullptr: also the d2 bits of the closest point seen other than the best (runner-up,
// min'd into *second_out: wave-uniform like best)
__device__ __forceinline__ unsigned long long coop_scan_box(const GridDev &g, float qx, float qy,
                                                            float qz, float r,
                                                            unsigned long long best, unsigned lane,
                                                            float *margin, unsigned *second_out = nullptr,
                                                            float pad = 0.f) {
    const float fx = (qx - g.ox) * g.inv_h, fy = (qy - g.oy) * g.inv_h, fz = (qz - g.oz) * g.inv_h;
    const float rc = r * g.inv_h + g.slack;
    const float big = 4.0e6f;
    const int x0 = (int) floorf(fminf(fmaxf(fx - rc, -big), big));
    const int x1 = (int) floorf(fminf(fmaxf(fx + rc, -big), big));
    const int y0 = (int) floorf(fminf(fmaxf(fy - rc, -big), big));
    const int y1 = (int) floorf(fminf(fmaxf(fy + rc, -big), big));
    const int z0 = (int) floorf(fminf(fmaxf(fz - rc, -big), big));
    const int z1 = (int) floorf(fminf(fmaxf(fz + rc, -big), big));
    const float mx = fminf(fx - (float) x0, (float) (x1 + 1) - fx);
    const float my = fminf(fy - (float) y0, (float) (y1 + 1) - fy);
    const float mz = fminf(fz - (float) z0, (float) (z1 + 1) - fz);
    *margin = (fminf(mx, fminf(my, mz)) - g.slack) * g.h;
    const int xa = max(x0, 0), xb = min(x1, g.nx - 1);
    const int ya = max(y0, 0), yb = min(y1, g.ny - 1);
    const int za = max(z0, 0), zb = min(z1, g.nz - 1);
    if (xa > xb || ya > yb || za > zb) return best;
    const int cy = (int) floorf(fminf(fmaxf(fy, -big), big));
    const int cz = (int) floorf(fminf(fmaxf(fz, -big), big));
    const int wy = yb - ya + 1;
    const int nrows = wy * (zb - za + 1);
    const float bd2 = __uint_as_float((unsigned) (best >> 32));
    // best distance (+ the room asked for above it, for the runner-up bound), in cells
    const float Rb = (__builtin_amdgcn_sqrtf(bd2) + pad) * g.inv_h * 1.00001f;
    unsigned long long mine = best;
    unsigned mine2 = 0x7F800000u;  // this lane's runner-up (second_out)
    for (int k0 = 0; k0 < nrows; k0 += 64) {
        // lanes resolve up to 64 rows at once
        const int k = k0 + (int) lane;
        unsigned s = 0, e = 0;
        if (k < nrows) {
            const int yy = ya + k % wy, zz = za + k / wy;
            const float ry = yy > cy ? (float) yy - fy : (yy < cy ? fy - (float) (yy + 1) : 0.f);
            const float rz = zz > cz ? (float) zz - fz : (zz < cz ? fz - (float) (zz + 1) : 0.f);
            // Only the part of the row inside ball(q, sqrt(best)) can hold a closer point: with
            // rho the (y,z) distance of the row, a point of the row closer than Rb has
            // |x - qx| <= sqrt(Rb^2 - (rho - slack)^2) <= sqrt(Rb^2 - rho^2 + 2 slack (Rb + slack))
            // (cell units; Rb inflated by 1e-5 against the approximate square roots).
            const float rho2 = ry * ry + rz * rz;
            const float lim = Rb + g.slack;
            if (!(rho2 > lim * lim)) {
                const float hx2 = fmaxf(Rb * Rb - rho2 + 2.f * g.slack * lim, 0.f);
                const float hx = __builtin_amdgcn_sqrtf(hx2) * 1.00001f + g.slack;
                const int xl = max(xa, (int) floorf(fminf(fmaxf(fx - hx, -big), big)));
                const int xh = min(xb, (int) floorf(fminf(fmaxf(fx + hx, -big), big)));
                if (xl <= xh) {
                    const size_t base = ((size_t) zz * g.ny + yy) * g.nx;
                    s = ldc(g.cell_start, base + xl);
                    e = ldc(g.cell_start, base + xh + 1);
                }
            }
        }
        unsigned long long rows = __ballot(e > s);
        while (rows) {
            const int rl = __ffsll((long long) rows) - 1;
            rows &= rows - 1;
            const unsigned rs = rl_u(s, rl), re = rl_u(e, rl);
            for (unsigned j = rs + lane; j < re; j += 128u) {
                const float4 t0 = ldp(g.pts, j);
                const float4 t1 = ldp(g.pts, j + 64u < re ? j + 64u : j);
                const unsigned long long a = make_key(canon_d2(qx, qy, qz, t0), __float_as_uint(t0.w));
                const unsigned long long b = make_key(canon_d2(qx, qy, qz, t1), __float_as_uint(t1.w));
                const unsigned long long m = a < b ? a : b;
                if (second_out) {  // (wave-uniform branch)
                    const unsigned long long hi = a < b ? b : a;
                    if (m < mine) {
                        mine2 = min(mine2, min((unsigned) (mine >> 32), hi != m ? (unsigned) (hi >> 32) : 0x7F800000u));
                    } else {
                        if (m != mine) mine2 = min(mine2, (unsigned) (m >> 32));
                        if (hi != mine && hi != m) mine2 = min(mine2, (unsigned) (hi >> 32));
                    }
                }
                mine = m < mine ? m : mine;
            }
        }
    }
    const unsigned long long all = wave_min_u64(mine);
    if (second_out) {
        // the wave's runner-up: the lanes' own runner-ups, and the bests of the lanes that do not hold the winner
        unsigned v = mine != all ? min(mine2, (unsigned) (mine >> 32)) : mine2;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = min(v, (unsigned) __shfl_xor((int) v, off));
        *second_out = min(*second_out, v);
    }
    return all;
}

// ------------------------------------------------- fused ICP statistics
// The search kernel ends with every lane holding its query (under the current pose), its
// match and d2 -- exactly what the statistics of the ICP step are summed from (wm_icp.hip:
// n, sum p, sum q, sum q p^T, sum d2 | GN: n, sum p, A^T A, J^T r, sum d2; + the number of
// points this rank handled).  Summing them here deletes a 40 MB stream and a launch per iteration.
//
// (the wave reduction by recursive halving: wm_wave.hpp)

// this lane's terms of the iteration's sums (same arithmetic as k_icp_stats, wm_icp.hip)
// a[17] counts the queries this rank handled; its fraction (units of 2^-24) counts those whose match
// CHANGED in this search -- what the host decides by whether the next searches can be certified instead
// (wm_icp_align).  Exact in a double, and never carrying into the integer part, because at most 2^23
// queries report: beyond that size only every (IcpDevState::changed_mask + 1)-th does (changed_mask_for,
// wm_icp_step.hpp) and the solve scales the count back up.
constexpr double kChangedUnit = 1.0 / 16777216.0;
template <int STATS>
__device__ __forceinline__ void icp_terms(double (&a)[kAcc], bool mine, bool matched, float qx, float qy, float qz,
                                          float bqx, float bqy, float bqz, float d2, bool changed = false) {
#pragma unroll
    for (int k = 0; k < kAcc; ++k) a[k] = 0.0;
    if (mine) {
        a[17] = changed ? 1.0 + kChangedUnit : 1.0;
        if (matched) {
            const double px = qx, py = qy, pz = qz, tx = bqx, ty = bqy, tz = bqz;
            a[0] = 1.0;
            a[1] = px;
            a[2] = py;
            a[3] = pz;
            if constexpr (STATS == WM_ICP_SVD) {
                a[4] = tx;
                a[5] = ty;
                a[6] = tz;
                a[7] = tx * px;
                a[8] = tx * py;
                a[9] = tx * pz;
                a[10] = ty * px;
                a[11] = ty * py;
                a[12] = ty * pz;
                a[13] = tz * px;
                a[14] = tz * py;
                a[15] = tz * pz;
            } else {
                const double rx = px - tx, ry = py - ty, rz = pz - tz;
                a[4] = py * py + pz * pz;
                a[5] = -px * py;
                a[6] = -px * pz;
                a[7] = px * px + pz * pz;
                a[8] = -py * pz;
                a[9] = px * px + py * py;
                a[10] = rx;
                a[11] = ry;
                a[12] = rz;
                a[13] = py * rz - pz * ry;
                a[14] = pz * rx - px * rz;
                a[15] = px * ry - py * rx;
            }
            a[16] = (double) d2;
        }
    }
}

// One lane per query: a certified radius search over a ladder of uniform grids
// (cell size x2 per level).
//   r <- the distance, under the NEW pose, to the point this query matched in the previous
//        iteration (a real candidate, so an upper bound of the answer); half a fine cell when
//        there is none;
//   repeat: scan ball(q, min(r, sqrt(best))) inside the box of cells covering [q - r, q + r]^3,
//           on the finest level whose cell is >= lane_lf * r (0.2 r: many short rows, each cut
//           to the ball's chord); certified when best <= margin(box) or the box already covers
//           max_corr; otherwise r <- best (something was found: the next scan is certain to
//           certify) or 2 r.
// Scans up to r_light (12 fine cells: all but ~1e-3 of the queries of a typical pair) run in the
// query's own lane (scan_box).  Longer ones are handed to the whole wavefront, one query at a
// time (coop_scan_box), seeded with the radius its Morton neighbour needed.
// The radius, level and pruning choices change the work, never the result.
// One wavefront per workgroup (finest dispatch granularity, smallest tail; workgroups of 4 / 5 /
// 10 waves cost the search 7 / 30 / 55 %).  STATS < 0: search only.  STATS = WM_ICP_SVD /
// WM_ICP_GN6: the wave also reduces the ICP statistics of its 64 queries to ONE row of `partials`
// ([gridDim.x][kAcc]).  BAL: the wave pools its lanes' candidate trips (balanced walk, above);
// otherwise every lane walks its own (kept for targets of 2^26 points and more, and for comparison).
template <int STATS, bool BAL, bool COST = false, int RC = kBalRowChunk, int WPE = 5>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
    k_nn_grid(const LevelsDev *__restrict__ lv, const float4 *__restrict__ src, unsigned n,
              IcpDevState *__restrict__ st, float thr_d2, unsigned long long *__restrict__ keys,
              float4 *__restrict__ match_pt, const float4 *__restrict__ tgt_orig,
              float r_light_cells, float lane_lf, float coop_lf, float r0_cells, unsigned xcd_chunk,
              double *__restrict__ partials, unsigned *__restrict__ cost_out,
              unsigned long long *__restrict__ phase_out, long long *__restrict__ bins) {
    // (bins != nullptr: the wave's sums are ADDED into the iteration's bins -- exact integer limbs, any order:
    // wm_bins.hpp -- instead of being stored as a row of `partials` for k_reduce_rows to add up)
    // (the wave's life is a chain of memory round trips; the three streams of its chunk -- source
    // point, previous key, previous match -- need nothing but the block number for their addresses
    // and are requested before the state is looked at.  Before the first search keys / match_pt hold
    // nothing meaningful and are not looked at.)
    const bool early = ((xcd_chunk >> 29) & 1u) != 0u;
    const bool rev = (xcd_chunk >> 31) != 0u;  // experiment: hand the queries out back to front
    const unsigned chunk_sz = xcd_chunk & 0x0FFFFFFFu;
    const bool nt = ((xcd_chunk >> 28) & 1u) != 0u;  // non-temporal result stores
    const bool walk_filter = ((xcd_chunk >> 30) & 1u) != 0u;  // see balanced_walk
    const unsigned bidx = rev ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned row = chunk_sz ? xcd_remap_chunked(bidx, chunk_sz) : xcd_remap(bidx, gridDim.x);
    const unsigned i = row * 64u + lane;
    const bool active = i < n;
    float4 p_e = make_float4(0.f, 0.f, 0.f, 0.f), tp_e = p_e;
    unsigned long long prev_e = ~0ull;
    if (early) {
        const unsigned ic = min(i, n - 1u);
        p_e = src[ic];
        prev_e = keys[ic];
        tp_e = match_pt[ic];
    }
    if (st->done) return;
    unsigned cost = 0;
    // developer (COST): shader-clock cycles of this wave's phases, added into phase_out[8] by lane 0:
    // [0] walk, [1] rounds, [2] walks, [3] prologue, [4] pass loop, [5] cooperative phase + stores,
    // [6] statistics tail, [7] waves
    unsigned long long prof[3] = {0ull, 0ull, 0ull};
    const unsigned long long prof_start = COST ? clock64() : 0ull;
    // per wave: [run][lane] = each lane's pending runs (lane scan), or the pooled trip list (balanced walk)
    __shared__ uint2 s_runs[BAL ? 1 : kRowChunk * 64];
    __shared__ BalHolder<BAL> s_hold;
    const int L = lv->n;
    const int L_levels = L;
    float hl[kMaxLevels];  // the levels' cell sizes (wave-uniform)
#pragma unroll
    for (int k = 0; k < kMaxLevels; ++k) hl[k] = lv->g[k < L ? k : 0].h;
    const float h0 = hl[0];
    const float rmax = sqrtf(thr_d2) * 1.0001f + 1e-6f;
    const float r_light = r_light_cells * h0;  // larger radii go to the cooperative path
    const bool have_prev = st->have_prev != 0;  // wave-uniform
    float qx = 0.f, qy = 0.f, qz = 0.f, r = 0.f;
    float bqx = 0.f, bqy = 0.f, bqz = 0.f;  // coordinates of the previous iteration's match
    unsigned long long best = make_key(thr_d2, kNoIdx);
    unsigned long long seeded = best;  // the key the search started from
    bool heavy = false;
    bool mine = active;
    // the query and its seed (previous key + previous match coordinates) are three
    // independent streams: all three loads are issued before the first use -- one memory round
    // trip, not three dependent ones
    unsigned long long prev = ~0ull;
    float4 tp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        const float4 p = early ? p_e : src[i];
        if (have_prev) {
            prev = early ? prev_e : keys[i];
            tp = early ? tp_e : match_pt[i];
        }
        xform(st->Tf, p, qx, qy, qz);
        // sharded registration: only the rank owning this x-slab handles the point
        if (st->slab_on && !(qx >= st->slab_lo && qx < st->slab_hi)) mine = false;
    }
    // a point of another rank's slab is left untouched: its key / match keep whatever this
    // rank last found for it (still a valid candidate if it ever comes back)
    if (mine) {
        r = r0_cells * h0;
        if (prev != ~0ull) {
            const unsigned pidx = (unsigned) prev;
            r = rmax;  // nothing (close enough) to start from: the full radius
            if (pidx != kNoIdx) {
                // the point matched in the previous iteration is a real candidate: its
                // distance under the NEW pose is an upper bound of the new NN distance, so
                // one scan of ball(q, that distance) is certain to certify
                const float d2b = canon_d2(qx, qy, qz, tp);
                if (d2b <= thr_d2) {
                    best = seeded = make_key(d2b, pidx);
                    bqx = tp.x;
                    bqy = tp.y;
                    bqz = tp.z;
                    r = fmaxf(sqrtf(d2b) * 1.0001f + 1e-6f, 0.05f * h0);
                }
            }
        }
        r = fminf(r, rmax);
        heavy = r > r_light;
    }
    const unsigned long long prof_pro = COST ? clock64() : 0ull;
    if constexpr (BAL) {
        BalLds &L = s_hold.v;
        L.q[lane] = make_float4(qx, qy, qz, 0.f);
        // parked until the search is over (see BalLds)
        L.seeded[lane] = seeded;
        L.bq[0][lane] = bqx;
        L.bq[1][lane] = bqy;
        L.bq[2][lane] = bqz;
        asm volatile("" ::: "memory");
        bool live = mine && !heavy;
        for (int pass = 0; pass < 32 && __ballot(live) != 0ull; ++pass) {
            int l = 0;  // the finest level whose cell is >= lane_lf * r (cell sizes double from level to level)
#pragma unroll
            for (int k = 0; k < kMaxLevels - 1; ++k) l += (k < L_levels - 1 && hl[k] < lane_lf * r) ? 1 : 0;
            // all live lanes on one level (always, once the clouds are close): the level's
            // description comes through scalar loads into SGPRs instead of eleven VGPRs per lane
            // (one level at a time, lanes of the other levels working along, was slower: 77.6 vs
            // 73.4 us per launch)
            const unsigned long long lv_mask = __ballot(live);
            const int l0 = __builtin_amdgcn_readlane(l, __ffsll((long long) lv_mask) - 1);
            float margin;
            if (__ballot(live && l != l0) == 0ull) {
                const GridDev g = lv->g[l0];
                best = scan_box_bal<COST, RC>(g, live, qx, qy, qz, r, best, &margin, L, lane, have_prev, g.pts, cost, prof, walk_filter);
            } else {
                const GridDev g = lv->g[l];
                L.base[lane] = (unsigned long long) g.pts;
                best = scan_box_bal<COST, RC>(g, live, qx, qy, qz, r, best, &margin, L, lane, have_prev, nullptr, cost, prof, walk_filter);
            }
            if (live) {
                if constexpr (COST) cost += 1u << 24;
                const float bd2 = __uint_as_float((unsigned) (best >> 32));
                if (margin > 0.f && (bd2 <= margin * margin || thr_d2 <= margin * margin)) {
                    live = false;
                } else {
                    const float rn = ((unsigned) best != kNoIdx) ? sqrtf(bd2) * 1.0001f + 1e-6f : 2.0f * r;
                    r = fminf(fmaxf(rn, 1.25f * r), rmax);
                    heavy = r > r_light;
                    live = !heavy;
                }
            }
        }
    } else if (mine) {
        for (int pass = 0; !heavy && pass < 32; ++pass) {
            int l = 0;
            while (l < L - 1 && lv->g[l].h < lane_lf * r) ++l;
            const GridDev g = lv->g[l];
            float margin;
            best = scan_box<COST>(g, qx, qy, qz, r, best, &margin, s_runs, lane, have_prev, cost);
            if constexpr (COST) cost += 1u << 24;  // passes
            const float bd2 = __uint_as_float((unsigned) (best >> 32));
            if (margin > 0.f && (bd2 <= margin * margin || thr_d2 <= margin * margin)) break;
            // not certified: the radius must GROW (a query sitting on a cell face can have a
            // non-positive margin however small its neighbour distance is)
            const float rn = ((unsigned) best != kNoIdx) ? sqrtf(bd2) * 1.0001f + 1e-6f : 2.0f * r;
            r = fminf(fmaxf(rn, 1.25f * r), rmax);
            heavy = r > r_light;
        }
    }
    const unsigned long long prof_pass = COST ? clock64() : 0ull;
    // ---- cooperative phase: the wave takes its heavy queries one at a time
    unsigned long long todo = __ballot(heavy);
    const unsigned n_heavy = __popcll(todo);
    float seed = 0.f;  // radius the previous heavy query of this wave ended with
    while (todo) {
        const int sl = __ffsll((long long) todo) - 1;
        todo &= todo - 1;
        const float ux = rl_f(qx, sl), uy = rl_f(qy, sl), uz = rl_f(qz, sl);
        float ur = rl_f(r, sl);
        unsigned long long ub = ((unsigned long long) rl_u((unsigned) (best >> 32), sl) << 32) |
                                rl_u((unsigned) best, sl);
        if ((unsigned) ub == kNoIdx && seed > ur) ur = fminf(seed, rmax);  // neighbour's radius
        for (int pass = 0; pass < 64; ++pass) {
            int l = 0;
            while (l < L - 1 && lv->g[l].h < coop_lf * ur) ++l;
            const GridDev g = lv->g[l];
            float margin;
            ub = coop_scan_box(g, ux, uy, uz, ur, ub, lane, &margin);
            const float bd2 = __uint_as_float((unsigned) (ub >> 32));
            if (margin > 0.f && (bd2 <= margin * margin || thr_d2 <= margin * margin)) break;
            if (ur >= rmax) break;
            const float rn = ((unsigned) ub != kNoIdx) ? sqrtf(bd2) * 1.0001f + 1e-6f : 2.0f * ur;
            ur = fminf(fmaxf(rn, 1.25f * ur), rmax);
        }
        seed = ((unsigned) ub != kNoIdx) ? 1.25f * sqrtf(__uint_as_float((unsigned) (ub >> 32))) : ur;
        if ((int) lane == sl) best = ub;
    }
    // (the lane number and the query index are formed again here, from mbcnt: kept from the top of
    // the kernel they -- and the LDS addresses derived from them -- were three registers the
    // compiler spilled to scratch, 16 MB of extra writes and reads per launch)
    const unsigned lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const unsigned i_e = BAL ? row * 64u + lane_e : i;
    if constexpr (BAL) {
        asm volatile("" ::: "memory");
        BalLds &L = s_hold.v;
        seeded = L.seeded[lane_e];
        bqx = L.bq[0][lane_e];
        bqy = L.bq[1][lane_e];
        bqz = L.bq[2][lane_e];
    }
    if (mine) {
        st_u64(&keys[i_e], best, nt);
        // the match's coordinates ride along for the statistics kernel and for the next
        // iteration's seed; a new winner's are read from the caller-ordered target copy
        // (its key carries the original index)
        if (best != seeded && (unsigned) best != kNoIdx) {
            const f4v c = ((gp_f4) tgt_orig)[(unsigned) best];
            bqx = c.x;
            bqy = c.y;
            bqz = c.z;
        }
        // (.w: the match's index, for k_nn_cert.)  A query that KEPT its match -- the seed's key, formed from match_pt[i]
        // itself under this pose, is still the best -- finds match_pt[i] already holding these very coordinates and
        // this index: no store (16 of the 24 result bytes of most queries once the clouds are close)
        if (!(best == seeded && (unsigned) best != kNoIdx))
            st_f4(&match_pt[i_e], bqx, bqy, bqz, __uint_as_float((unsigned) best), nt);
    }
    if (lane == 0 && n_heavy) atomicAdd(&st->queue_count[1], n_heavy);  // stats only
    const unsigned long long prof_store = COST ? clock64() : 0ull;
    if constexpr (COST) {
        if (active) cost_out[i_e] = mine ? (cost | (heavy ? 0x80000000u : 0u)) : 0u;
    }
    if constexpr (STATS >= 0) {
        double a[kAcc];
        icp_terms<STATS>(a, mine, (unsigned) best != kNoIdx, qx, qy, qz, bqx, bqy, bqz,
                         __uint_as_float((unsigned) (best >> 32)),
                         (unsigned) best != (unsigned) seeded && (i_e & st->changed_mask) == 0u);
        acc_halve<kAcc, 32>(a, lane);
        const int comp = acc_comp_of_lane(lane);
        if (comp >= 0) {
            if (bins) bins_add(bins, row % (unsigned) kBinCount, (unsigned) comp, a[0]);
            else st_f64(&partials[(size_t) row * kAcc + comp], a[0], nt);
        }
    }
    if constexpr (COST) {
        if (lane_e == 0 && phase_out) {
            const unsigned long long t_end = clock64();
            atomicAdd(&phase_out[0], prof[0]);
            atomicAdd(&phase_out[1], prof[1]);
            atomicAdd(&phase_out[2], prof[2]);
            atomicAdd(&phase_out[3], prof_pro - prof_start);
            atomicAdd(&phase_out[4], prof_pass - prof_pro);
            atomicAdd(&phase_out[5], prof_store - prof_pass);
            atomicAdd(&phase_out[6], t_end - prof_store);
            atomicAdd(&phase_out[7], 1ull);
        }
    }
}

// ------------------------------------------------- certified correspondences (late iterations)
// Once the clouds are nearly aligned an ICP step moves a source point by far less than the spacing of
// the target, and almost every query keeps its neighbour.  That can be PROVED per query without a
// search: the last search of the query left, besides its match m, the query's position and a lower
// bound s on its distance, there, to every target point other than m (runner-up tracking, `Bound`
// above).  If now, with disp the distance moved since,
//     |q - m| < s - disp        (with float-rounding cushions)
// then every other point is strictly farther than m: m is the exact nearest neighbour, ties
// included, and the query is SETTLED by three stream reads (source point, match, position + bound:
// 48 B) and NO write: its key keeps the match's index, and the distance in it is brought up to date
// once, when the registration ends (k_fix_keys).  Queries that fail the test are searched with the
// pooled walk of k_nn_grid, pruning with min(runner-up, best + pad) so that the bound they leave is
// worth something (pad = pad_mul x the size of the last step, at most pad_frac x the seed distance).
// Same keys, bit for bit, as a full search of every query.
//
// A workgroup of four waves handles 4 x NB x 64 consecutive (Morton-ordered) queries:
//   phase 1  every wave: NB batches of 64 -- certificate test, the settled queries' terms of the
//            iteration's sums (two batches added lane by lane, then a wave reduction); the unsettled
//            ones are listed, in query order, in the wave's LDS (the first 64 with their data);
//   phase 2  the unsettled queries of the whole workgroup in chunks of 64, chunk c by wave c mod 4:
//            the search, stores, sums.  Once aligned a workgroup has a handful of them: ONE wave
//            runs one chunk instead of four waves running one each -- the kernel is bound by
//            instruction issue, and a chunk costs the same whether 3 or 60 of its lanes are live;
//   the four waves' sums are added in wave order: one row of partial sums per workgroup (a 1M cloud
//   leaves 984 rows: the solve kernel adds them itself, no row-reduction launch).
// bounds_valid = 0: no usable bounds (the previous iteration was searched by k_nn_grid): every query
// is searched and leaves its bound.
constexpr int kCertWaves = 4;

// ---- the RESIDENT form of the certificate kernel (LATE = true): the late iterations of one registration
// in ONE launch.  Every workgroup keeps its 4 x NB x 64 queries from iteration to iteration (their three
// streams -- source point, match, position + bound -- are requested again while the solver works: L2 /
// Infinity Cache hits that cost no time of their own), the
// iteration's sums meet in device memory (one row per workgroup, written through; a ticket per workgroup),
// and ONE extra workgroup -- the solver, a kernel of its own on a second stream -- adds the rows in a fixed order,
// runs the solve and PCL's stopping rules (icp_apply_stats: what k_reduce_solve runs), publishes the
// iteration's record to the host and hands the new pose to the workers through a 64-byte slot.  What a
// launched certified iteration pays around its ~6 us of work -- two kernel boundaries, 48 MB of streams,
// the dispatch of 4 000 waves -- is gone.  Every workgroup has to be resident at once (checked on the
// host against the kernel's occupancy and the device's budget of resident workgroups); every wait gives
// up after kLateGuardTicks and sets `abandoned`, after which everybody leaves and the host continues with
// launched iterations from the state the solver wrote back.  No agent-scope fences anywhere (an XCD-wide
// L2 write-back each): rows, pose and counters are written through / read at agent scope.
// (every word that is polled or hammered sits in a 128-byte line of its own, and the word the thousand
// workers wait on exists sixteen times: a worker that has delivered its row looks at copy (workgroup mod
// 16) -- a thousand pollers of ONE line keep its memory channel so busy that the ticket atomics and row
// stores of the workgroups still working queue up behind them)
constexpr int kLateGenCopies = 16;
struct LateCtl {            // device memory
    unsigned ticket;        // rows delivered so far (monotonic over the iterations of one launch)
    unsigned pad0[31];
    unsigned abandoned;     // a wait timed out somewhere: everybody leaves
    unsigned pad1[31];
    float bc[2][16];        // by parity of the iteration: Tf[12], step size, flags (bit 0: stop), 2 spare
    struct {
        unsigned gen;       // iterations whose result the solver has handed out (0xFFFFFFFF: leave, a wait gave up)
        unsigned pad[31];
    } g[kLateGenCopies];
};
static_assert(sizeof(LateCtl) == 384 + 128 * kLateGenCopies, "layout");
struct LateArgs {
    LateCtl *ctl;
    unsigned long long *pub;     // pinned: the iterations' records (as k_reduce_solve writes them)
    int pub_slots;
    unsigned long long *h_exit;  // pinned: [exit_seq : 32 | reason : 8 | iterations done inside : 24], written last
    unsigned exit_seq;
    float stop_unsettled;        // leave when an iteration had to search more than this share of the queries ...
    float stop_disp;             // ... or a step moved the points by more than this (metres)
    int max_inside;              // ... or after this many iterations
    unsigned long long *dbg;     // developer (WM_LATE_DEBUG): 4 wall-clock stamps per iteration from the solver
    unsigned long long *dbg_w;   // ... and 8 per WORKER for iteration dbg_li
    unsigned dbg_li;
};
constexpr unsigned long long kLateGuardTicks = 20000000ull;  // 0.2 s of the 100 MHz wall clock
constexpr int kLateRow = 20;  // doubles per workgroup row: the kAcc sums, the queries it searched, one spare
enum { kLateDone = 1, kLatePolicy = 2, kLateAbandoned = 3, kLateBudget = 4 };

__device__ __forceinline__ unsigned ld_agent_u32(const unsigned *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the solver workgroup of the resident kernel: see above
struct LateSolverLds {
    IcpDevState st;
    double part[12][kLateRow];
    double tot[kLateRow];
    unsigned go, flags;
};
__global__ void __launch_bounds__(64 * kCertWaves) __attribute__((amdgpu_waves_per_eu(4, 4)))
    k_late_solver(const double *partials, unsigned workers, IcpDevState *st, LateArgs la) {
    // (a kernel of its own, on a second stream beside the workers': its f64 solve and its 28 loads in flight per
    // thread would otherwise set the register allocation of the search loop.  One workgroup, and no bigger than
    // a worker's in threads / registers / LDS: it fits wherever a worker fits)
    __shared__ LateSolverLds S;
    LateCtl *ctl = la.ctl;
    constexpr unsigned kWords = sizeof(IcpDevState) / 4;
    for (unsigned w = threadIdx.x; w < kWords; w += 64u * kCertWaves)
        reinterpret_cast<unsigned *>(&S.st)[w] = reinterpret_cast<const unsigned *>(st)[w];
    __syncthreads();
    unsigned reason = 0, inside = 0;
    if (S.st.done) reason = kLateDone;  // (queued behind a `done`: nothing to do -- the workers have left too)
    for (unsigned li = 0; reason == 0u; ++li) {
        // ---- all rows of this iteration in?
        if (threadIdx.x < 64u) {
            const unsigned want = (li + 1u) * workers;
            const unsigned long long t0 = wall_clock64();
            bool ok = true;
            for (;;) {
                if ((int) (ld_agent_u32(&ctl->ticket) - want) >= 0) break;
                if (ld_agent_u32(&ctl->abandoned) != 0u || wall_clock64() - t0 > kLateGuardTicks) {
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (threadIdx.x == 0) S.go = ok ? 1u : 0u;
            if (threadIdx.x == 0 && la.dbg && li < 64u) la.dbg[li * 4u + 0u] = wall_clock64();  // all rows in
        }
        __syncthreads();
        if (!S.go) {
            if (threadIdx.x == 0) __hip_atomic_store(&ctl->abandoned, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            reason = kLateAbandoned;
        } else {
            // ---- the rows, in a fixed order: thread (r, c) adds rows r, r + 12, ... of column c (28 loads in
            // flight per thread: three round trips for a thousand rows), one thread per column adds the 12
            const unsigned c = threadIdx.x % (unsigned) kLateRow, r = threadIdx.x / (unsigned) kLateRow;
            if (r < 12u) {
                double acc = 0.0;
                constexpr int U = 28;
                for (unsigned b = r; b < workers; b += 12u * U) {
                    double v[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const unsigned bb = b + 12u * (unsigned) u;
                        v[u] = bb < workers ? __hip_atomic_load(partials + (size_t) bb * kLateRow + c, __ATOMIC_RELAXED,
                                                                __HIP_MEMORY_SCOPE_AGENT)
                                            : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) acc += v[u];
                }
                S.part[r][c] = acc;
            }
            __syncthreads();
            if (threadIdx.x < (unsigned) kLateRow) {
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < 12; ++k) t += S.part[k][threadIdx.x];
                S.tot[threadIdx.x] = t;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                if (la.dbg && li < 64u) la.dbg[li * 4u + 1u] = wall_clock64();  // rows added
                double a[kAcc], ex[kStatsLen];
#pragma unroll
                for (int k = 0; k < kAcc; ++k) a[k] = S.tot[k];
                expand_stats(S.st.mode, a, ex, S.st.changed_mask);
                S.st.local_handled = ex[kStatsLen - 1];
#pragma unroll
                for (int k = 0; k < kStatsLen; ++k) S.st.stats[k] = ex[k];
                icp_apply_stats(&S.st, ex, (long long) S.tot[kAcc]);
                unsigned fl = 0;
                if (S.st.done) fl = kLateDone;
                else if ((li > 0u || S.st.frac_unsettled < 0.999f) && S.st.frac_unsettled > la.stop_unsettled) fl = kLatePolicy;
                else if (S.st.step_disp > la.stop_disp) fl = kLatePolicy;
                else if ((int) (li + 1u) >= la.max_inside) fl = kLateBudget;
                S.flags = fl;
                if (la.dbg && li < 64u) la.dbg[li * 4u + 2u] = wall_clock64();  // solved
                if (la.pub) {  // (the record k_reduce_solve publishes: same layout)
                    const unsigned f_ch = (unsigned) (fminf(fmaxf(S.st.frac_changed, 0.f), 1.f) * 65535.f + 0.5f);
                    const unsigned f_un = (unsigned) (fminf(fmaxf(S.st.frac_unsettled, 0.f), 1.f) * 65535.f + 0.5f);
                    const unsigned long long w = ((unsigned long long) ((unsigned) S.st.iter & 0xFFFFu) << 48) |
                                                 ((unsigned long long) (__float_as_uint(S.st.step_disp) >> 16) << 32) |
                                                 ((unsigned long long) f_ch << 16) | (unsigned long long) f_un;
                    if (S.st.iter >= 1 && S.st.iter <= la.pub_slots)
                        __hip_atomic_store(la.pub + S.st.iter, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(la.pub, S.st.done ? (1ull | ((unsigned long long) (unsigned) S.st.iter << 1)) : 0ull,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            __syncthreads();
            reason = S.flags;
            inside = li + 1u;
        }
        // ---- the pose of the next iteration (or the word to leave) for the workers: data, wait, then the number
        if (threadIdx.x < 16u) {
            const unsigned t = threadIdx.x;
            const float v = t < 12u ? S.st.Tf[t] : (t == 12u ? S.st.step_disp : (t == 13u ? __uint_as_float(reason) : 0.f));
            __hip_atomic_store(&ctl->bc[(li + 1u) & 1u][t], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x < (unsigned) kLateGenCopies)
            __hip_atomic_store(&ctl->g[threadIdx.x].gen, reason == (unsigned) kLateAbandoned ? ~0u : li + 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0 && la.dbg && li < 64u) la.dbg[li * 4u + 3u] = wall_clock64();  // handed out
    }
    // ---- the state goes back to memory for the kernels behind this one; the host learns how it ended
    __syncthreads();
    for (unsigned w = threadIdx.x; w < kWords; w += 64u * kCertWaves)
        reinterpret_cast<unsigned *>(st)[w] = reinterpret_cast<const unsigned *>(&S.st)[w];
    if (threadIdx.x == 0 && la.h_exit)
        __hip_atomic_store(la.h_exit, ((unsigned long long) la.exit_seq << 32) | ((unsigned long long) (reason & 0xFFu) << 24) |
                                          (unsigned long long) (inside & 0xFFFFFFu),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// (P1: developer timing experiment, WRONG results -- phase 2 compiled OUT: what the certificate phase costs in a kernel
// whose registers, scalar registers and LDS are not set by the searches; launched for bounds-valid launches only)
template <int STATS, int NB, int RC, bool LATE = false, bool P1 = false>
__global__ void __launch_bounds__(64 * kCertWaves) __attribute__((amdgpu_waves_per_eu(P1 ? 6 : 4, P1 ? 6 : 4)))
    k_nn_cert(const LevelsDev *__restrict__ lv, const float4 *__restrict__ src, unsigned n,
              IcpDevState *__restrict__ st, float thr_d2, unsigned long long *__restrict__ keys,
              float4 *__restrict__ match_pt, float4 *__restrict__ bound, const float4 *__restrict__ tgt_orig,
              float r_light_cells, float lane_lf, float coop_lf, float r0_cells, unsigned xflags,
              double *__restrict__ partials, int bounds_valid, float pad_mul, float pad_frac,
              unsigned *__restrict__ uns_count, unsigned long long *__restrict__ prof_out, LateArgs la,
              long long *__restrict__ bins) {
    // (bins != nullptr, launched form only: the workgroup's sums and its count of searched queries are ADDED into
    // the iteration's bins -- exact integer limbs, any order: wm_bins.hpp -- instead of stored as a row of `partials`)
    // (a wave's life is a chain of memory round trips: the phase's three streams are requested before
    // anything else is looked at -- their addresses need nothing but the block number)
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (unsigned) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const bool nt = ((xflags >> 28) & 1u) != 0u;  // non-temporal result stores (st_f4 ...)
    const unsigned dbg_skip = (xflags >> 26) & 3u;  // developer timing experiment (WRONG results): 1 = no phase 2, 2 = phase 2 without its scans
    const unsigned row = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned gbase = row * (64u * NB * kCertWaves);  // the workgroup's first query
    const unsigned base = gbase + wave * (64u * NB);        // the wave's
    float4 p[NB], mp[NB], rf[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const unsigned i = min(base + (unsigned) j * 64u + lane, n - 1u);
        p[j] = src[i];
        mp[j] = match_pt[i];  // (meaningless before the first search, and then not looked at)
        rf[j] = bound[i];
    }
    if (!LATE && st->done) return;  // (uniform over the workgroup)
    // LATE: the three streams of the later iterations come through buffer loads the compiler cannot hoist out
    // of the iteration loop (kept in registers across the searches they would be spilled to scratch: the
    // searches need every register); the match and the bound at agent scope (sc1), past this compute unit's
    // L1 -- the workgroup's own searches of the previous iteration rewrote some of them
    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void *) src, 0, LATE ? n * 16u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_mp = __builtin_amdgcn_make_buffer_rsrc((void *) match_pt, 0, LATE ? n * 16u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_rf = __builtin_amdgcn_make_buffer_rsrc((void *) bound, 0, LATE ? n * 16u : 0u, 0x00020000);
    // developer (prof_out): shader-clock stamps of wave 0 of every 256th workgroup, 16 per sample
    // (the phases between them: scripts/dev/dev_cert_prof.py); nothing is recorded otherwise
    const bool stamp_on = prof_out != nullptr && (blockIdx.x & 255u) == 0u && wave == 0u;
    unsigned long long pt[12];
#define WM_STAMP(k) do { if (stamp_on) pt[k] = clock64(); } while (0)
#pragma unroll
    for (int k = 0; k < 12; ++k) pt[k] = 0ull;
    WM_STAMP(0);
    __shared__ BalLds s_L[kCertWaves];
    __shared__ unsigned s_second[kCertWaves][64];
    __shared__ float4 s_win[kCertWaves][64];
    __shared__ unsigned short s_list[kCertWaves][64 * NB];
    __shared__ unsigned s_cnt[kCertWaves];
    __shared__ double s_rows[kCertWaves][kAcc];
    __shared__ float s_bc[20];  // LATE: this iteration's pose, step size, flags, [16] = a wait gave up
    if constexpr (LATE) {
        if (st->done) return;  // (a launch queued behind a `done`; the solver tells the host)
    }
    BalLds &L = s_L[wave];
    if constexpr (!P1) s_win[wave][lane] = make_float4(0.f, 0.f, 0.f, __uint_as_float(kNoIdx));  // (no winner recorded)
    const int Ln = lv->n;
    float hl[kMaxLevels];  // the levels' cell sizes (wave-uniform: scalar registers)
#pragma unroll
    for (int k = 0; k < kMaxLevels; ++k) hl[k] = lv->g[k < Ln ? k : 0].h;
    const GridDev g0 = lv->g[0];  // (what the late searches scan: fetched with the rest of the state, not when first needed)
    const float h0 = hl[0];
    const float rmax = sqrtf(thr_d2) * 1.0001f + 1e-6f;
    const float r_light = r_light_cells * h0;
    const bool have_prev = st->have_prev != 0;
    // sharded registration: this rank handles the queries whose transformed x lies in its slab (a query
    // it does not own is skipped: no test, no search, nothing stored -- whatever this rank knew about
    // it stays consistent for the day it comes back)
    const bool slab_on = st->slab_on != 0;
    const float slab_lo = st->slab_lo, slab_hi = st->slab_hi;
    const unsigned changed_mask = st->changed_mask;
    const int comp = acc_comp_of_lane(lane);
    WM_STAMP(1);  // state in
    // (LATE: one trip per iteration of the registration; otherwise one trip)
    for (unsigned li = 0;; ++li) {
    // ---- this iteration's pose, step size, and whether bounds exist
    if constexpr (LATE) {
        if (li > 0u) {
            // (requested BEFORE the wait for the solver: they arrive while it works)
            typedef unsigned u4v __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const unsigned i = min(base + (unsigned) j * 64u + lane, n - 1u);
                const u4v a = __builtin_amdgcn_raw_buffer_load_b128(rs_src, i * 16u, 0, 0);
                const u4v b = __builtin_amdgcn_raw_buffer_load_b128(rs_mp, i * 16u, 0, 16);
                const u4v c = __builtin_amdgcn_raw_buffer_load_b128(rs_rf, i * 16u, 0, 16);
                p[j] = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
                mp[j] = make_float4(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w));
                rf[j] = make_float4(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z), __uint_as_float(c.w));
            }
            if (wave == 0u) {  // wave 0 waits for the solver's word (one request per look), then fetches the slot
                // (the solver needs ~9 us from the last row to its word: a first long nap, then a look every ~0.5 us)
                const unsigned *my_gen = &la.ctl->g[blockIdx.x & (unsigned) (kLateGenCopies - 1)].gen;
                const unsigned long long t0 = wall_clock64();
                bool ok = true;
                __builtin_amdgcn_s_sleep(100);
                for (;;) {
                    const unsigned g = ld_agent_u32(my_gen);
                    if (g == ~0u || wall_clock64() - t0 > kLateGuardTicks) {
                        ok = false;
                        break;
                    }
                    if ((int) (g - li) >= 0) break;
                    __builtin_amdgcn_s_sleep(20);
                }
                if (lane < 16u)
                    s_bc[lane] = __uint_as_float(ld_agent_u32(reinterpret_cast<const unsigned *>(&la.ctl->bc[li & 1u][lane])));
                if (lane == 16u) s_bc[16] = ok ? 0.f : 1.f;
                if (!ok && lane == 0u) __hip_atomic_store(&la.ctl->abandoned, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (threadIdx.x < 17u) {
            s_bc[threadIdx.x] = threadIdx.x < 12u ? st->Tf[threadIdx.x] : (threadIdx.x == 12u ? st->step_disp : 0.f);
        }
        __syncthreads();
        if (s_bc[16] != 0.f || __float_as_uint(s_bc[13]) != 0u) return;  // (uniform: gave up, or told to leave)
    }
#define WM_WSTAMP(k)                                                                                            \
    do {                                                                                                        \
        if constexpr (LATE)                                                                                     \
            if (la.dbg_w && li == la.dbg_li && threadIdx.x == 0) la.dbg_w[(size_t) blockIdx.x * 8u + (k)] = wall_clock64(); \
    } while (0)
    WM_WSTAMP(0);  // pose in
    // (one launch per iteration: the pose is read where it is used, as before; resident: from this iteration's slot)
    float Tl_loc[12];
    if constexpr (LATE) {
#pragma unroll
        for (int k = 0; k < 12; ++k)
            Tl_loc[k] = __uint_as_float((unsigned) __builtin_amdgcn_readfirstlane((int) __float_as_uint(s_bc[k])));
    }
    const float *Tl = LATE ? Tl_loc : st->Tf;
    const float step_now = LATE ? __uint_as_float((unsigned) __builtin_amdgcn_readfirstlane((int) __float_as_uint(s_bc[12])))
                                : st->step_disp;
    const bool valid = (LATE && li > 0u) || (bounds_valid != 0 && have_prev);
    // room a search leaves above its result for the runner-up bound: a few of the last step's sizes
    // (what the following steps will add up to while the registration converges)
    const float pad_room = have_prev ? pad_mul * step_now : 0.f;
    double rowacc = 0.0;
    unsigned n_uns = 0;    // (wave-uniform)
    // ---- phase 1: the certificate
    {
        double acc[kAcc];
#pragma unroll
        for (int k = 0; k < kAcc; ++k) acc[k] = 0.0;
        bool any = false;
        if (stamp_on) {  // (when the first batch's three loads have landed)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pt[2] = clock64();
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const unsigned i = base + (unsigned) j * 64u + lane;
            const bool act = i < n;
            bool settled = false, owned = act;
            float qx = 0.f, qy = 0.f, qz = 0.f, d2 = 0.f;
            if (act) {
                xform(Tl, p[j], qx, qy, qz);
                if (slab_on && !(qx >= slab_lo && qx < slab_hi)) owned = false;
            }
            if (owned && valid) {
                // where the query is now, how far that is from where its bound was taken, and how far
                // its match is: no stores -- a settled query costs three stream reads
                const float ex = qx - rf[j].x, ey = qy - rf[j].y, ez = qz - rf[j].z;
                // (v_sqrt_f32, 1 ulp: the comparison carries 1e-4 relative + 1e-6 m of cushion on either side;
                // the library sqrtf is a twenty-instruction sequence, and this phase is bound by issue)
                const float disp = __builtin_amdgcn_sqrtf(ex * ex + ey * ey + ez * ez);
                const unsigned idx = __float_as_uint(mp[j].w);
                d2 = canon_d2(qx, qy, qz, mp[j]);
                settled = idx != kNoIdx && d2 <= thr_d2 &&
                          __builtin_amdgcn_sqrtf(d2) * 1.0001f + 1e-6f < rf[j].w - disp * 1.0001f - 1e-6f;
            }
            if constexpr (STATS >= 0) {
                // the settled queries' terms: two batches are added lane by lane, then one wave reduction
                // (all four at once needs 36 more live registers than the kernel has: 92 B of scratch per lane)
                if ((j & 1) == 0) any = false;
                any = any || settled;
                if constexpr (STATS == WM_ICP_SVD) {
                    // (the first batch of a pair assigns, the second accumulates with fused multiply-adds:
                    // half the f64 instructions of forming the terms and adding them)
                    const double m = settled ? 1.0 : 0.0;
                    const double px = settled ? (double) qx : 0.0, py = settled ? (double) qy : 0.0,
                                 pz = settled ? (double) qz : 0.0;
                    const double tx = settled ? (double) mp[j].x : 0.0, ty = settled ? (double) mp[j].y : 0.0,
                                 tz = settled ? (double) mp[j].z : 0.0;
                    const double dd = settled ? (double) d2 : 0.0;
                    if ((j & 1) == 0) {
                        acc[0] = m;
                        acc[1] = px;
                        acc[2] = py;
                        acc[3] = pz;
                        acc[4] = tx;
                        acc[5] = ty;
                        acc[6] = tz;
                        acc[7] = tx * px;
                        acc[8] = tx * py;
                        acc[9] = tx * pz;
                        acc[10] = ty * px;
                        acc[11] = ty * py;
                        acc[12] = ty * pz;
                        acc[13] = tz * px;
                        acc[14] = tz * py;
                        acc[15] = tz * pz;
                        acc[16] = dd;
                        acc[17] = m;
                    } else {
                        acc[0] += m;
                        acc[1] += px;
                        acc[2] += py;
                        acc[3] += pz;
                        acc[4] += tx;
                        acc[5] += ty;
                        acc[6] += tz;
                        acc[7] = fma(tx, px, acc[7]);
                        acc[8] = fma(tx, py, acc[8]);
                        acc[9] = fma(tx, pz, acc[9]);
                        acc[10] = fma(ty, px, acc[10]);
                        acc[11] = fma(ty, py, acc[11]);
                        acc[12] = fma(ty, pz, acc[12]);
                        acc[13] = fma(tz, px, acc[13]);
                        acc[14] = fma(tz, py, acc[14]);
                        acc[15] = fma(tz, pz, acc[15]);
                        acc[16] += dd;
                        acc[17] += m;
                    }
                } else {
                    double a[kAcc];
                    icp_terms<STATS>(a, settled, settled, qx, qy, qz, mp[j].x, mp[j].y, mp[j].z, d2);
#pragma unroll
                    for (int k = 0; k < kAcc; ++k) acc[k] = ((j & 1) == 0 ? 0.0 : acc[k]) + a[k];
                }
                if ((j & 1) == 1 || j == NB - 1) {
                    if (__ballot(any) != 0ull) {
                        acc_halve<kAcc, 32>(acc, lane);
                        rowacc += comp >= 0 ? acc[0] : 0.0;
                    }
                }
            }
            const bool uns = owned && !settled;
            const unsigned long long umask = __ballot(uns);
            if (!P1 && uns) {
                const unsigned before = __builtin_amdgcn_mbcnt_hi((unsigned) (umask >> 32),
                                                                  __builtin_amdgcn_mbcnt_lo((unsigned) umask, 0u));
                const unsigned e = n_uns + before;
                s_list[wave][e] = (unsigned short) ((unsigned) j * 64u + lane);
                // the wave's first 64 are parked (pose applied, match) where its pooled walk will keep
                // its list: phase 2 starts without another round trip to memory
                if (valid && e < 64u) {
                    float4 *park = reinterpret_cast<float4 *>(L.items);
                    park[2u * e] = make_float4(qx, qy, qz, 0.f);
                    park[2u * e + 1u] = mp[j];
                }
            }
            n_uns += (unsigned) __popcll(umask);
        }
    }
    if (lane == 0) s_cnt[wave] = n_uns;
    WM_STAMP(3);  // phase 1 done
    __syncthreads();
    WM_WSTAMP(1);  // phase 1 done (all waves)
    // ---- phase 2: what is left in the workgroup, 64 queries at a time, chunk c by wave c mod 4
    unsigned cum[kCertWaves + 1];
    cum[0] = 0;
#pragma unroll
    for (int w = 0; w < kCertWaves; ++w) cum[w + 1] = cum[w] + s_cnt[w];
    unsigned U = cum[kCertWaves];
    if (uns_count && threadIdx.x == 0 && U) atomicAdd(&uns_count[blockIdx.x & 63u], U);  // developer statistics
    // how many queries this launch had to search: the solve kernel hands it to the host (one atomic per
    // workgroup, spread over 64 words)
    // (with bins the count is one of their components)
    if (!LATE && !bins && threadIdx.x == 0 && U) atomicAdd(&st->cert_unsettled[blockIdx.x & 63u], U);
    const unsigned U_searched = U;
    if (dbg_skip == 1u && valid) U = 0;
    if constexpr (P1) U = 0;
    const unsigned nchunks = (U + 63u) / 64u;
    unsigned cost = 0;
    unsigned long long prof[3] = {0ull, 0ull, 0ull};
    // The wave's first chunk is gathered from the four waves' parked entries BEFORE any wave scans (a
    // scan overwrites its wave's parking area), and set down again in the wave's own area after the
    // barrier: query (pose applied), its index, its match.
    if constexpr (!P1) {
        float gx = 0.f, gy = 0.f, gz = 0.f;
        float4 gtp = make_float4(0.f, 0.f, 0.f, __uint_as_float(kNoIdx));
        unsigned gi = kNoIdx;
        const unsigned e = wave * 64u + lane;
        if (wave < nchunks && e < U) {
            const unsigned w = (e >= cum[1] ? 1u : 0u) + (e >= cum[2] ? 1u : 0u) + (e >= cum[3] ? 1u : 0u);
            const unsigned k = e - (w == 0u ? cum[0] : (w == 1u ? cum[1] : (w == 2u ? cum[2] : cum[3])));
            gi = gbase + w * (64u * NB) + (unsigned) s_list[w][k];
            if (valid && k < 64u) {
                const float4 *park = reinterpret_cast<const float4 *>(s_L[w].items);
                const float4 a = park[2u * k];
                gtp = park[2u * k + 1u];
                gx = a.x;
                gy = a.y;
                gz = a.z;
            } else {
                const float4 p1 = src[gi];
                if (have_prev) gtp = match_pt[gi];
                xform(Tl, p1, gx, gy, gz);
            }
        }
        __syncthreads();
        float4 *own = reinterpret_cast<float4 *>(L.items);
        own[2u * lane] = make_float4(gx, gy, gz, __uint_as_float(gi));
        own[2u * lane + 1u] = gtp;
    }
    for (unsigned c = wave; c < nchunks; c += kCertWaves) {
        bool mine;
        unsigned i;
        float qx = 0.f, qy = 0.f, qz = 0.f, r = 0.f, pad = 0.f;
        float bqx = 0.f, bqy = 0.f, bqz = 0.f;
        float4 tp = make_float4(0.f, 0.f, 0.f, __uint_as_float(kNoIdx));
        bool from_mem = false;
        if (c == wave) {
            const float4 *own = reinterpret_cast<const float4 *>(L.items);
            const float4 a = own[2u * lane];
            tp = own[2u * lane + 1u];
            i = __float_as_uint(a.w);
            mine = i != kNoIdx;
            qx = a.x;
            qy = a.y;
            qz = a.z;
            __builtin_amdgcn_wave_barrier();  // (read before the walk reuses this LDS)
        } else {
            const unsigned e = c * 64u + lane;
            mine = e < U;
            i = 0;
            if (mine) {
                const unsigned w = (e >= cum[1] ? 1u : 0u) + (e >= cum[2] ? 1u : 0u) + (e >= cum[3] ? 1u : 0u);
                const unsigned k = e - (w == 0u ? cum[0] : (w == 1u ? cum[1] : (w == 2u ? cum[2] : cum[3])));
                i = gbase + w * (64u * NB) + (unsigned) s_list[w][k];
            }
            from_mem = mine;
        }
        unsigned long long best = make_key(thr_d2, kNoIdx);
        unsigned long long seeded = best;
        bool heavy = false;
        if (from_mem) {
            const float4 p1 = src[i];
            if (have_prev) tp = match_pt[i];
            xform(Tl, p1, qx, qy, qz);
        }
        if (mine) {
            r = r0_cells * h0;
            if (have_prev) {
                const unsigned pidx = __float_as_uint(tp.w);
                r = rmax;
                if (pidx != kNoIdx) {
                    const float d2b = canon_d2(qx, qy, qz, tp);
                    if (d2b <= thr_d2) {
                        best = seeded = make_key(d2b, pidx);
                        bqx = tp.x;
                        bqy = tp.y;
                        bqz = tp.z;
                        const float sd = sqrtf(d2b);
                        pad = fminf(pad_room, pad_frac * sd);
                        r = fmaxf(sd * 1.0001f + 1e-6f, 0.05f * h0) + pad;
                    }
                }
            }
            r = fminf(r, rmax);
            heavy = r > r_light;
        }
        if (c == wave) WM_STAMP(4);  // first chunk: seeds ready
        Bound bnd;
        bnd.second = 0x7F800000u;
        bnd.pad = pad;
        bnd.lds = s_second[wave];
        bnd.win = s_win[wave];
        bnd.ok = true;
        float margin_last = 0.f;
        L.q[lane] = make_float4(qx, qy, qz, 0.f);
        L.seeded[lane] = seeded;
        L.bq[0][lane] = bqx;
        L.bq[1][lane] = bqy;
        L.bq[2][lane] = bqz;
        asm volatile("" ::: "memory");
        bool live = mine && !heavy && !(dbg_skip == 2u && valid);
        for (int pass = 0; pass < 32 && __ballot(live) != 0ull; ++pass) {
            int l = 0;  // the finest level whose cell is >= lane_lf * r (cell sizes double from level to level)
#pragma unroll
            for (int k = 0; k < kMaxLevels - 1; ++k) l += (k < Ln - 1 && hl[k] < lane_lf * r) ? 1 : 0;
            const unsigned long long lv_mask = __ballot(live);
            const int l0 = __builtin_amdgcn_readlane(l, __ffsll((long long) lv_mask) - 1);
            float margin;
            if (__ballot(live && l != l0) == 0ull) {
                const GridDev g = l0 == 0 ? g0 : lv->g[l0];
                if (!scan_box_rows(g, live, qx, qy, qz, r, best, &margin, L, lane, &bnd))
                    best = scan_box_bal<false, RC, true>(g, live, qx, qy, qz, r, best, &margin, L, lane, have_prev,
                                                          g.pts, cost, prof, true, &bnd);
            } else {
                const GridDev g = lv->g[l];
                L.base[lane] = (unsigned long long) g.pts;
                best = scan_box_bal<false, RC, true>(g, live, qx, qy, qz, r, best, &margin, L, lane, have_prev,
                                                      nullptr, cost, prof, true, &bnd);
            }
            if (live) {
                const float bd2 = __uint_as_float((unsigned) (best >> 32));
                margin_last = margin;
                if (margin > 0.f && (bd2 <= margin * margin || thr_d2 <= margin * margin)) {
                    live = false;
                } else {
                    const float rn = ((unsigned) best != kNoIdx) ? sqrtf(bd2) * 1.0001f + 1e-6f + pad : 2.0f * r;
                    r = fminf(fmaxf(rn, 1.25f * r), rmax);
                    heavy = r > r_light;
                    live = !heavy;
                }
            }
        }
        if (c == wave) WM_STAMP(5);  // first chunk: pass loop done
        // cooperative phase for radii beyond r_light (k_nn_grid's), with the runner-up tracked as well: the
        // ball scanned is the query's whole search ball, so the bound is min(runner-up, margin of the last box)
        unsigned long long todo = __ballot(heavy);
        float seed = 0.f;
        while (todo) {
            const int sl = __ffsll((long long) todo) - 1;
            todo &= todo - 1;
            const float ux = rl_f(qx, sl), uy = rl_f(qy, sl), uz = rl_f(qz, sl);
            float ur = rl_f(r, sl);
            unsigned long long ub = ((unsigned long long) rl_u((unsigned) (best >> 32), sl) << 32) |
                                    rl_u((unsigned) best, sl);
            unsigned usec = 0x7F800000u;
            float umargin = 0.f;
            const float upad = rl_f(pad, sl);
            if ((unsigned) ub == kNoIdx && seed > ur) ur = fminf(seed, rmax);
            for (int pass = 0; pass < 64; ++pass) {
                int l = 0;
                while (l < Ln - 1 && lv->g[l].h < coop_lf * ur) ++l;
                const GridDev g = lv->g[l];
                float margin;
                ub = coop_scan_box(g, ux, uy, uz, ur, ub, lane, &margin, &usec, upad);
                umargin = margin;
                const float bd2 = __uint_as_float((unsigned) (ub >> 32));
                if (margin > 0.f && (bd2 <= margin * margin || thr_d2 <= margin * margin)) break;
                if (ur >= rmax) break;
                const float rn = ((unsigned) ub != kNoIdx) ? sqrtf(bd2) * 1.0001f + 1e-6f : 2.0f * ur;
                ur = fminf(fmaxf(rn, 1.25f * ur), rmax);
            }
            seed = ((unsigned) ub != kNoIdx) ? 1.25f * sqrtf(__uint_as_float((unsigned) (ub >> 32))) : ur;
            if ((int) lane == sl) {
                best = ub;
                // (a scan cuts its rows to the chord of ball(q, best at its entry + pad): what it skipped is
                // farther than that, hence farther than the final best + pad)
                bnd.second = usec;
                margin_last = umargin;
                heavy = false;
            }
        }
        asm volatile("" ::: "memory");
        seeded = L.seeded[lane];
        bqx = L.bq[0][lane];
        bqy = L.bq[1][lane];
        bqz = L.bq[2][lane];
        if (c == wave) WM_STAMP(6);  // first chunk: cooperative phase done
        if (mine) {
            st_u64(&keys[i], best, nt);
            if (best != seeded && (unsigned) best != kNoIdx) {
                // the new match's coordinates: left in LDS by the lane that found it (the tag says whether
                // the slot really is this point's), else from the caller-ordered target copy
                const float4 w = s_win[wave][lane];
                if (__float_as_uint(w.w) == (unsigned) best) {
                    bqx = w.x;
                    bqy = w.y;
                    bqz = w.z;
                } else {
                    const f4v cc = ((gp_f4) tgt_orig)[(unsigned) best];
                    bqx = cc.x;
                    bqy = cc.y;
                    bqz = cc.z;
                }
            }
            st_f4(&match_pt[i], bqx, bqy, bqz, __uint_as_float((unsigned) best), nt);
            // every point but the match is farther than: the runner-up seen, the radius pruned with, and
            // the faces of the last box scanned
            float s = 0.f;
            if (bnd.ok && !heavy && (unsigned) best != kNoIdx && margin_last > 0.f) {
                const float bd = sqrtf(__uint_as_float((unsigned) (best >> 32)));
                s = fminf(fminf(sqrtf(__uint_as_float(bnd.second)), bd + pad), margin_last) * 0.9999f - 1e-6f;
            }
            st_f4(&bound[i], qx, qy, qz, s, nt);  // ... seen from HERE
        }
        if (c == wave) WM_STAMP(7);  // first chunk: winners fetched, results stored
        if constexpr (STATS >= 0) {
            double a[kAcc];
            icp_terms<STATS>(a, mine, (unsigned) best != kNoIdx, qx, qy, qz, bqx, bqy, bqz,
                             __uint_as_float((unsigned) (best >> 32)),
                             (unsigned) best != (unsigned) seeded && (i & changed_mask) == 0u);
            acc_halve<kAcc, 32>(a, lane);
            rowacc += comp >= 0 ? a[0] : 0.0;
        }
        if (c == wave) WM_STAMP(8);  // first chunk: sums reduced
        __builtin_amdgcn_wave_barrier();
    }
    WM_STAMP(10);
    if constexpr (LATE) {
        WM_WSTAMP(2);  // wave 0's searches done
        // the four waves' sums in wave order -> the workgroup's row, written through; when the stores have
        // been performed, the ticket.  (Every wave first waits for its own result stores: the next
        // iteration's loads of the match and the bound, by other waves, come behind the barrier.)
        if (comp >= 0) s_rows[wave][comp] = rowacc;
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        WM_WSTAMP(3);  // all waves' searches done, result stores performed
        if (threadIdx.x == 0 && la.dbg_w && li == la.dbg_li) la.dbg_w[(size_t) blockIdx.x * 8u + 6u] = U_searched;
        if (threadIdx.x < (unsigned) kAcc + 1u) {
            double t;
            if (threadIdx.x < (unsigned) kAcc) {
                t = s_rows[0][threadIdx.x];
#pragma unroll
                for (int w = 1; w < kCertWaves; ++w) t += s_rows[w][threadIdx.x];
            } else {
                t = (double) U_searched;
            }
            __hip_atomic_store(partials + (size_t) row * kLateRow + threadIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        WM_WSTAMP(4);  // row stored
        if (threadIdx.x == 0) (void) __hip_atomic_fetch_add(&la.ctl->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        WM_WSTAMP(5);  // ticket drawn
    } else if constexpr (STATS >= 0) {
        // the four waves' sums, added in wave order
        if (comp >= 0) s_rows[wave][comp] = rowacc;
        __syncthreads();
        if (bins) {
            if (threadIdx.x < (unsigned) kAcc) {
                double t = s_rows[0][threadIdx.x];
#pragma unroll
                for (int w = 1; w < kCertWaves; ++w) t += s_rows[w][threadIdx.x];
                bins_add(bins, row % (unsigned) kBinCount, threadIdx.x, t);
            } else if (threadIdx.x == (unsigned) kAcc && U_searched) {
                bins_add_count(bins, row % (unsigned) kBinCount, (unsigned) kAcc, (long long) U_searched);
            }
        } else if (threadIdx.x < (unsigned) kAcc) {
            double t = s_rows[0][threadIdx.x];
#pragma unroll
            for (int w = 1; w < kCertWaves; ++w) t += s_rows[w][threadIdx.x];
            st_f64(&partials[(size_t) row * kAcc + threadIdx.x], t, nt);
        }
    }
    if (stamp_on && lane == 0) {
        pt[11] = clock64();
        unsigned long long *o = prof_out + 16 * (blockIdx.x >> 8);
#pragma unroll
        for (int k = 0; k < 12; ++k) o[k] = pt[k];
        o[12] = U;
    }
    if constexpr (!LATE) break;
    }  // (iterations)
#undef WM_WSTAMP
#undef WM_STAMP
}

// After a registration whose last searches were certified: the settled queries' keys still carry the
// distance of their last real search.  Bring every key up to date with the pose of the last search
// (same arithmetic as the search: same bits as if every query had been searched).
__global__ void __launch_bounds__(kBlock)
    k_fix_keys(const float4 *__restrict__ src, unsigned n, const IcpDevState *__restrict__ st, float thr_d2,
               const float4 *__restrict__ match_pt, unsigned long long *__restrict__ keys) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = src[i], m = match_pt[i];
    const unsigned idx = __float_as_uint(m.w);
    float qx, qy, qz;
    xform(st->Tf_search, p, qx, qy, qz);
    keys[i] = idx == kNoIdx ? make_key(thr_d2, kNoIdx) : make_key(canon_d2(qx, qy, qz, m), idx);
}

int launch_fix_keys(wm_ctx *ctx, float thr_d2) {
    const unsigned n = (unsigned) ctx->n_src;
    if (n == 0) return WM_OK;
    hipLaunchKernelGGL(k_fix_keys, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream,
                       ctx->src_sorted.as<float4>(), n, ctx->d_state.as<IcpDevState>(), thr_d2,
                       ctx->match_pt.as<float4>(), ctx->keys.as<unsigned long long>());
    WM_HIP(ctx, hipGetLastError());
    return WM_OK;
}

// ----------------------------------------------------------- brute force
constexpr int kBruteTile = 1024;

__global__ void __launch_bounds__(kBlock) k_init_keys(unsigned long long *keys, unsigned n,
                                                       const IcpDevState *st, float thr_d2) {
    if (st->done) return;
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) keys[i] = make_key(thr_d2, kNoIdx);
}

// grid = (ceil(n / 256), splits); block y scans target slice y.  Every lane
// holds one (transformed) query; the target streams through a float4 LDS tile
// that all lanes read at the same address (LDS broadcast, no bank conflicts).
__global__ void __launch_bounds__(kBlock)
    k_nn_brute(const float4 *__restrict__ tgt, unsigned m, const float4 *__restrict__ src,
               unsigned n, const IcpDevState *__restrict__ st, float thr_d2,
               unsigned long long *__restrict__ keys) {
    if (st->done) return;
    __shared__ float4 tile[kBruteTile];
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    const unsigned per = (m + gridDim.y - 1) / gridDim.y;
    const unsigned m0 = blockIdx.y * per, m1 = min(m0 + per, m);
    float qx = 0, qy = 0, qz = 0;
    if (i < n) xform(st->Tf, src[i], qx, qy, qz);
    const bool mine = !(st->slab_on && !(qx >= st->slab_lo && qx < st->slab_hi));
    unsigned long long best = make_key(thr_d2, kNoIdx);
    for (unsigned t0 = m0; t0 < m1; t0 += kBruteTile) {
        const unsigned cnt = min((unsigned) kBruteTile, m1 - t0);
        __syncthreads();
        for (unsigned k = threadIdx.x; k < cnt; k += kBlock) tile[k] = tgt[t0 + k];
        __syncthreads();
        for (unsigned k = 0; k < cnt; ++k) {
            const float4 t = tile[k];
            // non-finite target points were packed as NaN: their key (0x7FC0....)
            // exceeds every finite threshold and never wins
            const unsigned long long key = make_key(canon_d2(qx, qy, qz, t), __float_as_uint(t.w));
            best = key < best ? key : best;
        }
    }
    if (i < n && mine && best < make_key(thr_d2, kNoIdx)) atomicMin(&keys[i], best);
}

// after the all-pairs search: coordinates of every match (the grid search tracks them itself)
__global__ void __launch_bounds__(kBlock)
    k_fill_match(const unsigned long long *__restrict__ keys, unsigned n,
                 const float4 *__restrict__ tgt, const IcpDevState *__restrict__ st,
                 float4 *__restrict__ match_pt) {
    if (st->done) return;
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const unsigned idx = (unsigned) keys[i];
    match_pt[i] = idx == kNoIdx ? make_float4(0.f, 0.f, 0.f, __uint_as_float(kNoIdx)) : tgt[idx];
}

// largest float whose value, compared as PCL does ((double) d2 > max_corr^2 ->
// reject), is still accepted
float threshold_d2(double max_corr) {
    const double m2 = max_corr * max_corr;
    if (!(m2 < 3.0e38)) return 3.0e38f;
    float f = (float) m2;
    if ((double) f > m2) f = nextafterf(f, 0.0f);
    return f;
}

// strict variant: accepted iff (double) d2 < max_corr^2 (estimateLUMold's gate)
float threshold_d2_strict(double max_corr) {
    const double m2 = max_corr * max_corr;
    if (!(m2 < 3.0e38)) return 3.0e38f;
    float f = (float) m2;
    if ((double) f >= m2) f = nextafterf(f, 0.0f);
    return f;
}

template <int STATS, bool BAL, bool COST = false, int RC = kBalRowChunk, int WPE = 5>
static void launch_nn_grid_t(wm_ctx *ctx, unsigned blocks, float thr_d2, unsigned xcd_chunk, long long *bins = nullptr) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nn_grid<STATS, BAL, COST, RC, WPE>), dim3(blocks), dim3(64), 0, ctx->stream,
                       ctx->d_levels.as<LevelsDev>(), ctx->src_sorted.as<float4>(), (unsigned) ctx->n_src,
                       ctx->d_state.as<IcpDevState>(), thr_d2, ctx->keys.as<unsigned long long>(),
                       ctx->match_pt.as<float4>(), ctx->tgt_orig.as<float4>(), ctx->tune_r_light,
                       ctx->tune_lane_lf, ctx->tune_coop_lf, ctx->tune_r0, xcd_chunk, ctx->partials.as<double>(),
                       ctx->cost_log.p ? ctx->cost_log.as<unsigned>() + (size_t) ctx->cost_log_iter * ctx->n_src : nullptr,
                       ctx->phase_log.p ? ctx->phase_log.as<unsigned long long>() + 8 * (size_t) ctx->cost_log_iter : nullptr, bins);
}

// stats_mode < 0: search only.  WM_ICP_SVD / WM_ICP_GN6: the search kernel also leaves the ICP
// statistics of this iteration as *rows_out rows of kAcc doubles in ctx->partials.
// use_bins (with a stats_mode): the sums go into the iteration's bins (wm_bins.hpp) -- *rows_out is 0 then, and the
// solve is launch_bins_solve
int launch_nn_grid(wm_ctx *ctx, float thr_d2, hipEvent_t ev0, hipEvent_t ev1, hipEvent_t ev2,
                   int stats_mode, unsigned *rows_out, bool use_bins) {
    const unsigned n = (unsigned) ctx->n_src;
    if (rows_out) *rows_out = 0;
    if (n == 0) return WM_OK;
    unsigned blocks = (n + 63u) / 64u;
    blocks = (blocks + 7u) & ~7u;  // xcd_remap needs a multiple of 8
    unsigned xcd_chunk = 0;
    if (ctx->tune_xcd_chunk > 0 && blocks >= 32u * (unsigned) ctx->tune_xcd_chunk) {
        // the chunked remap (big grids only: it pads the grid to a multiple of 8 chunks)
        xcd_chunk = (unsigned) ctx->tune_xcd_chunk;
        const unsigned m = 8u * xcd_chunk;
        blocks = (blocks + m - 1u) / m * m;
    }
    long long *bins = nullptr;
    if (stats_mode >= 0 && use_bins) {
        if (!ctx->bins.p) return WM_ERR_STATE;  // (the caller's loop made them ready: bins_ready, wm_icp.hip)
        bins = ctx->bins.as<long long>();
    } else if (stats_mode >= 0) {
        WM_HIP(ctx, ctx->partials.reserve((size_t) blocks * kAcc * sizeof(double)));
        if (rows_out) *rows_out = blocks;
    }
    xcd_chunk |= ctx->tune_xcd_reverse ? 0x80000000u : 0u;
    xcd_chunk |= ctx->tune_nn_walk_filter ? 0x40000000u : 0u;
    xcd_chunk |= ctx->tune_nn_early_loads ? 0x20000000u : 0u;
    xcd_chunk |= ctx->tune_nn_nt_stores ? 0x10000000u : 0u;
    // the balanced walk packs (lane, point offset) into 32 bits: targets below 2^26 points
    const bool bal = ctx->tune_nn_balanced && ctx->n_tgt_input < (1u << 26) - 8u;
    if (ev0) WM_HIP(ctx, hipEventRecord(ev0, ctx->stream));
    if (stats_mode == WM_ICP_SVD && ctx->cost_log.p && ctx->cost_log_iter < ctx->cost_log_cap) {
        if (bal) launch_nn_grid_t<WM_ICP_SVD, true, true>(ctx, blocks, thr_d2, xcd_chunk, bins);  // developer statistics
        else launch_nn_grid_t<WM_ICP_SVD, false, true>(ctx, blocks, thr_d2, xcd_chunk, bins);
        ctx->cost_log_iter++;
    } else if (stats_mode < 0) {
        if (bal) launch_nn_grid_t<-1, true>(ctx, blocks, thr_d2, xcd_chunk);
        else launch_nn_grid_t<-1, false>(ctx, blocks, thr_d2, xcd_chunk);
    } else if (stats_mode == WM_ICP_SVD) {
        // (developer: registers per wave of the production instantiation -- waves per SIMD 4 / 6 and rows per step 2)
        if (bal && ctx->tune_grid_variant == 1) launch_nn_grid_t<WM_ICP_SVD, true, false, 3, 4>(ctx, blocks, thr_d2, xcd_chunk, bins);
        else if (bal && ctx->tune_grid_variant == 2) launch_nn_grid_t<WM_ICP_SVD, true, false, 3, 6>(ctx, blocks, thr_d2, xcd_chunk, bins);
        else if (bal && ctx->tune_grid_variant == 3) launch_nn_grid_t<WM_ICP_SVD, true, false, 2, 5>(ctx, blocks, thr_d2, xcd_chunk, bins);
        else if (bal && ctx->tune_grid_variant == 4) launch_nn_grid_t<WM_ICP_SVD, true, false, 2, 6>(ctx, blocks, thr_d2, xcd_chunk, bins);
        else if (bal) launch_nn_grid_t<WM_ICP_SVD, true>(ctx, blocks, thr_d2, xcd_chunk, bins);
        else launch_nn_grid_t<WM_ICP_SVD, false>(ctx, blocks, thr_d2, xcd_chunk, bins);
    } else {
        if (bal) launch_nn_grid_t<WM_ICP_GN6, true>(ctx, blocks, thr_d2, xcd_chunk, bins);
        else launch_nn_grid_t<WM_ICP_GN6, false>(ctx, blocks, thr_d2, xcd_chunk, bins);
    }
    if (ev1) WM_HIP(ctx, hipEventRecord(ev1, ctx->stream));
    if (ev2) WM_HIP(ctx, hipEventRecord(ev2, ctx->stream));
    WM_HIP(ctx, hipGetLastError());
    return WM_OK;
}

template <int STATS, int NB, int RC>
static void launch_nn_cert_t(wm_ctx *ctx, unsigned blocks, float thr_d2, unsigned xcd_chunk, bool bounds_valid, long long *bins) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nn_cert<STATS, NB, RC, false>), dim3(blocks), dim3(64 * kCertWaves), 0, ctx->stream,
                       ctx->d_levels.as<LevelsDev>(), ctx->src_sorted.as<float4>(), (unsigned) ctx->n_src,
                       ctx->d_state.as<IcpDevState>(), thr_d2, ctx->keys.as<unsigned long long>(),
                       ctx->match_pt.as<float4>(), ctx->nn_bound.as<float4>(), ctx->tgt_orig.as<float4>(),
                       ctx->tune_r_light, ctx->tune_lane_lf, ctx->tune_coop_lf, ctx->tune_r0, xcd_chunk,
                       ctx->partials.as<double>(), bounds_valid ? 1 : 0, ctx->tune_cert_pad_mul,
                       ctx->tune_cert_pad_frac,
                       ctx->cert_count.p && ctx->cert_log_iter < ctx->cert_log_cap
                           ? ctx->cert_count.as<unsigned>() + 64 * (size_t) ctx->cert_log_iter : nullptr,
                       ctx->cert_prof.p && ctx->cert_log_iter < ctx->cert_log_cap
                           ? ctx->cert_prof.as<unsigned long long>() + 64 * (size_t) ctx->cert_log_iter : nullptr,
                       LateArgs{}, bins);
}

template <int NB, int RC>
static void launch_nn_cert_nb(wm_ctx *ctx, unsigned blocks, float thr_d2, unsigned xcd_chunk, bool bounds_valid,
                              int stats_mode, long long *bins) {
    if (stats_mode < 0) launch_nn_cert_t<-1, NB, RC>(ctx, blocks, thr_d2, xcd_chunk, bounds_valid, nullptr);
    else if (stats_mode == WM_ICP_SVD) launch_nn_cert_t<WM_ICP_SVD, NB, RC>(ctx, blocks, thr_d2, xcd_chunk, bounds_valid, bins);
    else launch_nn_cert_t<WM_ICP_GN6, NB, RC>(ctx, blocks, thr_d2, xcd_chunk, bounds_valid, bins);
}

// use_bins (with a stats_mode): sums and the searched-queries count go into the iteration's bins (wm_bins.hpp) --
// *rows_out is 0 then, and the solve is launch_bins_solve
int launch_nn_cert(wm_ctx *ctx, float thr_d2, hipEvent_t ev0, hipEvent_t ev1, hipEvent_t ev2, int stats_mode,
                   unsigned *rows_out, bool bounds_valid, bool use_bins) {
    const unsigned n = (unsigned) ctx->n_src;
    if (rows_out) *rows_out = 0;
    if (n == 0) return WM_OK;
    const int nb = ctx->tune_cert_nb == 2 ? 2 : 4;
    const unsigned per = 64u * (unsigned) nb * (unsigned) kCertWaves;
    unsigned blocks = (n + per - 1u) / per;
    blocks = (blocks + 7u) & ~7u;  // xcd_remap needs a multiple of 8
    WM_HIP(ctx, ctx->nn_bound.reserve(((size_t) n + 64) * sizeof(float4)));
    long long *bins = nullptr;
    if (stats_mode >= 0 && use_bins) {
        if (!ctx->bins.p) return WM_ERR_STATE;  // (the caller's loop made them ready: bins_ready, wm_icp.hip)
        bins = ctx->bins.as<long long>();
    } else if (stats_mode >= 0) {
        WM_HIP(ctx, ctx->partials.reserve((size_t) blocks * kAcc * sizeof(double)));
        if (rows_out) *rows_out = blocks;
    }
    const unsigned xflags = (ctx->tune_nn_nt_stores ? 0x10000000u : 0u) | (((unsigned) ctx->tune_cert_dbg_skip & 3u) << 26);
    if (ev0) WM_HIP(ctx, hipEventRecord(ev0, ctx->stream));
    if (ctx->tune_cert_dbg_skip == 3 && bounds_valid && stats_mode == WM_ICP_SVD && nb == 4) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nn_cert<WM_ICP_SVD, 4, 3, false, true>), dim3(blocks), dim3(64 * kCertWaves), 0, ctx->stream,
                           ctx->d_levels.as<LevelsDev>(), ctx->src_sorted.as<float4>(), (unsigned) ctx->n_src,
                           ctx->d_state.as<IcpDevState>(), thr_d2, ctx->keys.as<unsigned long long>(),
                           ctx->match_pt.as<float4>(), ctx->nn_bound.as<float4>(), ctx->tgt_orig.as<float4>(),
                           ctx->tune_r_light, ctx->tune_lane_lf, ctx->tune_coop_lf, ctx->tune_r0, xflags,
                           ctx->partials.as<double>(), 1, ctx->tune_cert_pad_mul, ctx->tune_cert_pad_frac,
                           (unsigned *) nullptr, (unsigned long long *) nullptr, LateArgs{}, bins);
    } else if (nb == 2) launch_nn_cert_nb<2, 3>(ctx, blocks, thr_d2, xflags, bounds_valid, stats_mode, bins);
    else launch_nn_cert_nb<4, 3>(ctx, blocks, thr_d2, xflags, bounds_valid, stats_mode, bins);
    if (ctx->cert_count.p && ctx->cert_log_iter < ctx->cert_log_cap) ctx->cert_log_iter++;
    if (ev1) WM_HIP(ctx, hipEventRecord(ev1, ctx->stream));
    if (ev2) WM_HIP(ctx, hipEventRecord(ev2, ctx->stream));
    WM_HIP(ctx, hipGetLastError());
    return WM_OK;
}

// ---- the resident form (k_nn_cert<.., LATE = true>)
constexpr int kLateNB = 4;
template <int STATS>
static const void *late_kernel() {
    return (const void *) k_nn_cert<STATS, kLateNB, 3, true>;
}

// workgroups of resident kernels (this one, GICP's evaluators) a device may hold at once, per process:
// resident kernels that each hold part of the GPU while waiting must never keep each other's remaining
// workgroups from starting
// (in 1/1024ths of the device: a kernel of nb workgroups of which `capacity` fit at once takes
// ceil(1024 nb / capacity) -- the kernels differ in what a workgroup occupies)
static std::atomic<int> g_resident[64];
int resident_admit(int device, int nb, int capacity) {
    if (device < 0 || device >= 64 || capacity <= 0 || nb > capacity) return 0;
    const int share = (int) (((long long) nb * 1024 + capacity - 1) / capacity);
    int cur = g_resident[device].load();
    while (cur + share <= 1024)
        if (g_resident[device].compare_exchange_weak(cur, cur + share)) return share;
    return 0;
}
void resident_release(int device, int share) {
    if (device >= 0 && device < 64 && share > 0) g_resident[device].fetch_sub(share);
}

size_t late_ctl_bytes() { return sizeof(LateCtl); }

// Can the late iterations of this align run in one resident launch?  (*blocks_out: its grid)
bool late_possible(wm_ctx *ctx, int stats_mode, unsigned *blocks_out) {
    const unsigned n = (unsigned) ctx->n_src;
    if (n == 0 || (stats_mode != WM_ICP_SVD && stats_mode != WM_ICP_GN6)) return false;
    if (ctx->late_capacity == 0) {  // first use: how many of its workgroups fit on the device at once?
        ctx->late_capacity = -1;
        int cus = 0, per_cu = 0, per_cu2 = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess) return false;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, late_kernel<WM_ICP_SVD>(), 64 * kCertWaves, 0) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu2, late_kernel<WM_ICP_GN6>(), 64 * kCertWaves, 0) != hipSuccess) {
            (void) hipGetLastError();
            return false;
        }
        ctx->late_capacity = cus * (per_cu < per_cu2 ? per_cu : per_cu2);
    }
    if (ctx->late_capacity <= 0) return false;
    const unsigned per = 64u * (unsigned) kLateNB * (unsigned) kCertWaves;
    unsigned workers = (n + per - 1u) / per;
    workers = (workers + 7u) & ~7u;  // xcd_remap needs a multiple of 8
    if ((int) workers + 1 > ctx->late_capacity) return false;  // (+ 1: the solver's workgroup)
    if (blocks_out) *blocks_out = workers;
    return true;
}

// Enqueue the resident kernel: iterations from the state's current one until done / the policy says leave
// / max_inside.  The caller holds `blocks` of the device's resident budget until the kernel has finished.
int launch_nn_late(wm_ctx *ctx, float thr_d2, int stats_mode, unsigned blocks, bool bounds_valid, unsigned exit_seq,
                   float stop_unsettled, float stop_disp, int max_inside) {
    const unsigned n = (unsigned) ctx->n_src;
    const unsigned workers = blocks;
    if (!ctx->side_stream || !ctx->ev_fork || !ctx->ev_join) return WM_ERR_STATE;
    WM_HIP(ctx, ctx->nn_bound.reserve(((size_t) n + 64) * sizeof(float4)));
    WM_HIP(ctx, ctx->partials.reserve((size_t) workers * kLateRow * sizeof(double)));
    WM_HIP(ctx, ctx->late_ctl.reserve(sizeof(LateCtl) + 64 * 4 * sizeof(unsigned long long)));
    if (!ctx->h_late) {
        WM_HIP(ctx, hipHostMalloc((void **) &ctx->h_late, 64, hipHostMallocDefault));
        *ctx->h_late = 0ull;
    }
    WM_HIP(ctx, hipMemsetAsync(ctx->late_ctl.p, 0, sizeof(LateCtl) + 64 * 4 * sizeof(unsigned long long), ctx->stream));
    LateArgs la;
    la.ctl = ctx->late_ctl.as<LateCtl>();
    la.pub = ctx->h_pub;
    la.pub_slots = ctx->h_pub_slots;
    la.h_exit = ctx->h_late;
    la.exit_seq = exit_seq;
    la.stop_unsettled = stop_unsettled;
    la.stop_disp = stop_disp;
    la.max_inside = max_inside;
    la.dbg = getenv("WM_LATE_DEBUG") ? (unsigned long long *) ((char *) ctx->late_ctl.p + sizeof(LateCtl)) : nullptr;
    la.dbg_w = nullptr;
    la.dbg_li = 0;
    if (la.dbg) {
        WM_HIP(ctx, ctx->cert_prof.reserve((size_t) workers * 8 * sizeof(unsigned long long)));
        WM_HIP(ctx, hipMemsetAsync(ctx->cert_prof.p, 0, (size_t) workers * 8 * sizeof(unsigned long long), ctx->stream));
        la.dbg_w = ctx->cert_prof.as<unsigned long long>();
        la.dbg_li = (unsigned) atoi(getenv("WM_LATE_DEBUG"));
    }
    const unsigned xflags = (ctx->tune_nn_nt_stores ? 0x10000000u : 0u);
    // the solver beside the workers, on the second stream: both start when what is on the main stream now is done
    WM_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    WM_HIP(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
    hipLaunchKernelGGL(k_late_solver, dim3(1), dim3(64 * kCertWaves), 0, ctx->side_stream, ctx->partials.as<double>(),
                       workers, ctx->d_state.as<IcpDevState>(), la);
    WM_HIP(ctx, hipGetLastError());
    WM_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->side_stream));
#define WM_LATE_LAUNCH(MODE)                                                                                            \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nn_cert<MODE, kLateNB, 3, true>), dim3(blocks), dim3(64 * kCertWaves), 0,          \
                       ctx->stream, ctx->d_levels.as<LevelsDev>(), ctx->src_sorted.as<float4>(), n,                       \
                       ctx->d_state.as<IcpDevState>(), thr_d2, ctx->keys.as<unsigned long long>(),                        \
                       ctx->match_pt.as<float4>(), ctx->nn_bound.as<float4>(), ctx->tgt_orig.as<float4>(),                \
                       ctx->tune_r_light, ctx->tune_lane_lf, ctx->tune_coop_lf, ctx->tune_r0, xflags,                     \
                       ctx->partials.as<double>(), bounds_valid ? 1 : 0, ctx->tune_cert_pad_mul, ctx->tune_cert_pad_frac, \
                       (unsigned *) nullptr, (unsigned long long *) nullptr, la, (long long *) nullptr)
    if (stats_mode == WM_ICP_SVD) WM_LATE_LAUNCH(WM_ICP_SVD);
    else WM_LATE_LAUNCH(WM_ICP_GN6);
#undef WM_LATE_LAUNCH
    WM_HIP(ctx, hipGetLastError());
    // (what follows on the main stream needs the state the solver writes back when it leaves)
    WM_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    return WM_OK;
}

int launch_nn_brute(wm_ctx *ctx, float thr_d2, hipEvent_t ev0, hipEvent_t ev1) {
    const unsigned n = (unsigned) ctx->n_src, m = (unsigned) ctx->n_tgt_input;
    if (n == 0) return WM_OK;
    const IcpDevState *st = ctx->d_state.as<IcpDevState>();
    unsigned long long *keys = ctx->keys.as<unsigned long long>();
    const unsigned bx = (n + kBlock - 1) / kBlock;
    unsigned splits = 1;
    if (m > 0) {
        // enough workgroups to fill 256 CUs several times over
        splits = (2048 + bx - 1) / bx;
        const unsigned max_splits = (m + kBruteTile - 1) / kBruteTile;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    if (ev0) WM_HIP(ctx, hipEventRecord(ev0, ctx->stream));
    hipLaunchKernelGGL(k_init_keys, dim3(bx), dim3(kBlock), 0, ctx->stream, keys, n, st, thr_d2);
    if (m > 0)
        hipLaunchKernelGGL(k_nn_brute, dim3(bx, splits), dim3(kBlock), 0, ctx->stream,
                           ctx->tgt_orig.as<float4>(), m, ctx->src_sorted.as<float4>(), n, st,
                           thr_d2, keys);
    hipLaunchKernelGGL(k_fill_match, dim3(bx), dim3(kBlock), 0, ctx->stream, keys, n,
                       ctx->tgt_orig.as<float4>(), st, ctx->match_pt.as<float4>());
    if (ev1) WM_HIP(ctx, hipEventRecord(ev1, ctx->stream));
    WM_HIP(ctx, hipGetLastError());
    return WM_OK;
}

}  // namespace wm
