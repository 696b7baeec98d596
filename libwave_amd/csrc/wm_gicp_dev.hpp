// wm_gicp_dev.hpp -- the device functions of GICP that more than one kernel is made of: the k-NN search of
// computeCovariances, the covariance of a neighbour list, the Mahalanobis matrix of a pair, the objective's terms of
// a pair and their double-double sums.  Used by wm_gicp.hip (one registration on the whole device) and
// wm_gicp_small.hip (one registration per workgroup, many per launch).  [PCL registration/impl/gicp.hpp]
#pragma once
#include "wm_internal.hpp"

namespace wm {

constexpr int kGicpAcc = 13;

__device__ __forceinline__ unsigned long long g_make_key(float d2, unsigned idx) {
    return ((unsigned long long) __float_as_uint(d2) << 32) | idx;
}
__device__ __forceinline__ float g_d2(float qx, float qy, float qz, const float4 &t) {
    const float dx = qx - t.x, dy = qy - t.y, dz = qz - t.z;
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// sorted insertion into an ascending register-resident list (drops the largest).  The list is sorted, so
// inserting is a shift: with c_j = (key < best[j]) -- false ... false true ... true --
//     new best[j] = c_{j-1} ? best[j-1] : (c_j ? key : best[j]),
// and for the distance word (the high one) alone that is the median of (best[j-1], key, best[j]).  One 64-bit
// compare, one v_med3_u32 and two selects per slot, going down the list so that best[j-1] is still the old one
// (the compare-and-swap chain this replaces compiled to two 64-bit compares, four selects and a move: 85
// instructions per candidate for the whole wave at K = 10, now 45).
__device__ __forceinline__ unsigned g_med3_u32(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <int K>
__device__ __forceinline__ void knn_insert(unsigned long long (&best)[K], unsigned long long key) {
    if (key >= best[K - 1]) return;
    const unsigned kh = (unsigned) (key >> 32), kl = (unsigned) key;
    bool c[K];
#pragma unroll
    for (int j = 0; j < K; ++j) c[j] = key < best[j];
#pragma unroll
    for (int j = K - 1; j >= 1; --j) {
        const unsigned ah = (unsigned) (best[j - 1] >> 32), al = (unsigned) best[j - 1];
        const unsigned bh = (unsigned) (best[j] >> 32), bl = (unsigned) best[j];
        const unsigned nh = g_med3_u32(ah, kh, bh);
        const unsigned nl = c[j - 1] ? al : (c[j] ? kl : bl);
        best[j] = ((unsigned long long) nh << 32) | nl;
    }
    best[0] = c[0] ? key : best[0];
}

// k nearest neighbours of q among the cell-sorted points of grid g: scan the box of cells
// covering ball(q, r); certified once the k-th distance is within the box margin.
//
// A pass resolves the box's rows (runs of x-adjacent cells = contiguous points) kKnnRows at a
// time -- their cell_start look-ups are issued together, unconditionally (a row outside the box
// reads row 0 and is given an empty run) -- pushes the non-empty runs into the lane's own column
// of `runs` (LDS) and then walks them in ONE flat loop, a candidate per trip: a lane moves on to its
// next run the moment its current one ends, so the wave makes max-over-lanes(candidates of a lane)
// trips rather than sum-over-rows(max-over-lanes(row length)).  (Same structure, and for the same
// reason, as the correspondence search's lane scan.)
constexpr int kKnnRows = 8;
#ifdef WM_COV_COUNT  // developer build: what the search does, summed over the launch (printed by launch_cov)
__device__ double g_knn_cnt[8];  // candidates, wave trips, batches (per lane), wave batches, passes (per lane), queries
#define WM_KNN_CNT(i, v) atomicAdd(&g_knn_cnt[i], (double) (v))
#else
#define WM_KNN_CNT(i, v) ((void) 0)
#endif
// A later pass (the box had to grow: the k-th distance found exceeded the first box's margin, the usual case at
// PCL's k = 10 on a grid of ~3 points per cell) does NOT start over: the list keeps what the smaller box gave, and
// only the SHELL between the two boxes is scanned -- the x-extensions of the rows the old box had, whole rows
// elsewhere, four rows (eight segments) per batch -- and of the shell only the rows the sphere of the k-th distance
// found so far can reach.  (Re-scanning the whole bigger box was 75 candidates through the sorted insertion per
// query instead of ~45; the insertion -- K compare-swaps for the whole wave per candidate -- is 85 % of k_gicp_cov.)
template <int K>
__device__ void knn_search(const GridDev &g, float qx, float qy, float qz, int k, float r0_cells,
                           unsigned long long (&best)[K], uint2 *runs, unsigned lane_col, unsigned col_stride,
                           float r_stop = 3.0e38f) {
    const float fx = (qx - g.ox) * g.inv_h, fy = (qy - g.oy) * g.inv_h, fz = (qz - g.oz) * g.inv_h;
    float r = r0_cells * g.h;
    const float rmax = (float) (g.nx + g.ny + g.nz + 3) * g.h;  // covers the whole grid
#pragma unroll
    for (int j = 0; j < K; ++j) best[j] = ~0ull;
    // the runs of one batch (s >= e: none) -> the lane's column of the LDS list -> ONE flat candidate loop
    auto scan_slots = [&](const unsigned (&rs)[kKnnRows], const unsigned (&re)[kKnnRows]) {
        int n_runs = 0;
#pragma unroll
        for (int u = 0; u < kKnnRows; ++u)
            if (re[u] > rs[u]) {
                runs[n_runs * col_stride + lane_col] = make_uint2(rs[u], re[u]);
                ++n_runs;
            }
#ifdef WM_COV_COUNT
        float my_c = 0.f, my_t = 0.f;
        WM_KNN_CNT(2, 1);
        WM_KNN_CNT(3, 1.0 / __popcll(__ballot(1)));
#endif
        if (n_runs == 0) return;
        // (the next candidate is fetched before the current one goes through the insertion: the load's way to L2
        // and back is as long as the insertion itself)
        int ri = 0;
        const uint2 r0 = runs[lane_col];
        unsigned j = r0.x, e = r0.y;
        float4 t = g.pts[j];
        for (;;) {
            bool more = true;
            if (++j == e) {
                more = ++ri < n_runs;
                if (more) {
                    const uint2 rn = runs[ri * col_stride + lane_col];
                    j = rn.x;
                    e = rn.y;
                }
            }
            const float4 tn = g.pts[more ? j : r0.x];  // (a lane at its end reads a line it has had already)
            knn_insert<K>(best, g_make_key(g_d2(qx, qy, qz, t), __float_as_uint(t.w)));
#ifdef WM_COV_COUNT
            my_c += 1.f;
            my_t += 1.f / (float) __popcll(__ballot(1));
            if (!more) {
                WM_KNN_CNT(0, my_c);
                WM_KNN_CNT(1, my_t);
            }
#endif
            if (!more) break;
            t = tn;
        }
    };
    int pxa = 1, pxb = 0, pya = 1, pyb = 0, pza = 1, pzb = 0;  // the box already scanned (clamped cells; none yet)
    for (int pass = 0; pass < 64; ++pass) {  // r at least x1.5 per pass: rmax is reached long before
        WM_KNN_CNT(4, 1);
        const float rc = r * g.inv_h + g.slack;
        const int x0 = (int) floorf(fx - rc), x1 = (int) floorf(fx + rc);
        const int y0 = (int) floorf(fy - rc), y1 = (int) floorf(fy + rc);
        const int z0 = (int) floorf(fz - rc), z1 = (int) floorf(fz + rc);
        const float mx = fminf(fx - (float) x0, (float) (x1 + 1) - fx);
        const float my = fminf(fy - (float) y0, (float) (y1 + 1) - fy);
        const float mz = fminf(fz - (float) z0, (float) (z1 + 1) - fz);
        const float margin = (fminf(mx, fminf(my, mz)) - g.slack) * g.h;
        const int xa = max(x0, 0), xb = min(x1, g.nx - 1);
        const int ya = max(y0, 0), yb = min(y1, g.ny - 1);
        const int za = max(z0, 0), zb = min(z1, g.nz - 1);
        const bool any = xa <= xb && ya <= yb && za <= zb;
        const bool have_prev = pxa <= pxb && pya <= pyb && pza <= pzb;
        int yy = ya, zz = any ? za : zb + 1;  // row cursor; zz > zb = past the last row
        if (!have_prev) {
            while (zz <= zb) {
                unsigned rs[kKnnRows], re[kKnnRows];
#pragma unroll
                for (int u = 0; u < kKnnRows; ++u) {
                    const bool live = zz <= zb;
                    const size_t base = ((size_t) (live ? zz : za) * g.ny + (live ? yy : ya)) * g.nx;
                    rs[u] = g.cell_start[base + xa];
                    re[u] = live ? g.cell_start[base + xb + 1] : 0u;  // dead row: e <= s
                    if (++yy > yb) {
                        yy = ya;
                        ++zz;
                    }
                }
                scan_slots(rs, re);
            }
        } else {
            // how far a point that still matters can be (cell units): the k-th distance found so far, if there is one
            unsigned long long kth0 = ~0ull;
#pragma unroll
            for (int j = 0; j < K; ++j)
                if (j == k - 1) kth0 = best[j];
            const float reach = kth0 != ~0ull ? sqrtf(__uint_as_float((unsigned) (kth0 >> 32))) * g.inv_h * 1.0001f + g.slack
                                              : 3.0e38f;
            const float reach2 = reach < 1.0e18f ? reach * reach : 3.0e38f;
            while (zz <= zb) {
                unsigned rs[kKnnRows], re[kKnnRows];
#pragma unroll
                for (int u = 0; u < kKnnRows; u += 2) {
                    const bool live = zz <= zb;
                    const int ry_ = live ? yy : ya, rz_ = live ? zz : za;
                    // the row's distance from the query in (y, z), cell units (0 inside the query's own row)
                    const float dy = fmaxf(fmaxf((float) ry_ - fy, fy - (float) (ry_ + 1)), 0.f);
                    const float dz = fmaxf(fmaxf((float) rz_ - fz, fz - (float) (rz_ + 1)), 0.f);
                    const bool reachable = live && !(dy * dy + dz * dz > reach2);
                    const bool old_row = ry_ >= pya && ry_ <= pyb && rz_ >= pza && rz_ <= pzb;
                    const size_t base = ((size_t) rz_ * g.ny + ry_) * g.nx;
                    // old row: [xa, pxa - 1] and [pxb + 1, xb]; new row: [xa, xb] and nothing
                    const int a0 = xa, a1 = old_row ? pxa - 1 : xb;
                    const int b0 = pxb + 1, b1 = xb;
                    const bool sa = reachable && a0 <= a1, sb = reachable && old_row && b0 <= b1;
                    rs[u] = sa ? g.cell_start[base + a0] : 0u;
                    re[u] = sa ? g.cell_start[base + a1 + 1] : 0u;
                    rs[u + 1] = sb ? g.cell_start[base + b0] : 0u;
                    re[u + 1] = sb ? g.cell_start[base + b1 + 1] : 0u;
                    if (++yy > yb) {
                        yy = ya;
                        ++zz;
                    }
                }
                scan_slots(rs, re);
            }
        }
        if (any) {
            pxa = xa, pxb = xb, pya = ya, pyb = yb, pza = za, pzb = zb;
        }
        unsigned long long kth = ~0ull;
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (j == k - 1) kth = best[j];
        const bool covers_all = x0 <= 0 && y0 <= 0 && z0 <= 0 && x1 >= g.nx - 1 && y1 >= g.ny - 1 &&
                                z1 >= g.nz - 1;
        if (covers_all) return;
        if (margin >= r_stop) return;  // (everything within r_stop of the query has been seen: the caller wants nothing farther)
        if (kth != ~0ull && margin > 0.f) {
            const float kd2 = __uint_as_float((unsigned) (kth >> 32));
            if (kd2 <= margin * margin) return;
            r = fmaxf(sqrtf(kd2) * 1.0001f + 1e-6f, 1.5f * r);  // one more pass certifies
        } else {
            r *= 2.0f;
        }
        r = fminf(r, rmax);
    }
}

// the covariance PCL's computeCovariances gives a point from its k nearest neighbours (best[0 .. k-1], keys of
// (d2, index); fetch(index) = the neighbour's coordinates): float products into double sums, 3x3 SVD, spectrum
// replaced by (1, 1, eps).  no_svd: developer timing experiment (the raw covariance goes out).
template <int K, class Fetch>
__device__ __forceinline__ void gicp_cov_of_list(const unsigned long long (&best)[K], int k, double eps, Fetch fetch,
                                                 double *__restrict__ out, bool no_svd = false) {
    double mean[3] = {0, 0, 0}, c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < K; ++j) {
        if (j < k && best[j] != ~0ull) {
            const float4 p = fetch((unsigned) best[j]);
            mean[0] += p.x;
            mean[1] += p.y;
            mean[2] += p.z;
            c[0] += __fmul_rn(p.x, p.x);  // float products, as `cov(0,0) += pt.x*pt.x`
            c[3] += __fmul_rn(p.y, p.x);
            c[4] += __fmul_rn(p.y, p.y);
            c[6] += __fmul_rn(p.z, p.x);
            c[7] += __fmul_rn(p.z, p.y);
            c[8] += __fmul_rn(p.z, p.z);
        }
    }
    const double kk = (double) k;
#pragma unroll
    for (int a = 0; a < 3; ++a) mean[a] /= kk;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
            if (b <= a) {
                c[a * 3 + b] /= kk;
                c[a * 3 + b] -= mean[a] * mean[b];
                c[b * 3 + a] = c[a * 3 + b];
            }
    double U[9], S[3], V[9];
    if (no_svd) {
#pragma unroll
        for (int a = 0; a < 9; ++a) out[a] = c[a];
        return;
    }
    svd3<false>(c, U, S, V);  // IEEE operations only: the oracle reproduces these matrices bit for bit
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            double s = 0;
#pragma unroll
            for (int j = 0; j < 3; ++j) s += (j == 2 ? eps : 1.0) * U[a * 3 + j] * U[b * 3 + j];
            out[a * 3 + b] = s;
        }
}

struct Mat3d {
    double m[9];
};

__device__ inline void inv3(const double *m, double *o) {
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8],
                 c02 = m[3] * m[7] - m[4] * m[6];
    const double id = 1.0 / (m[0] * c00 + m[1] * c01 + m[2] * c02);
    o[0] = c00 * id;
    o[1] = (m[2] * m[7] - m[1] * m[8]) * id;
    o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id;
    o[4] = (m[0] * m[8] - m[2] * m[6]) * id;
    o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id;
    o[7] = (m[1] * m[6] - m[0] * m[7]) * id;
    o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// mahalanobis_[i] = (R C1_i R^T + C2_j)^-1
__device__ __forceinline__ void gicp_mahal_of(const double *__restrict__ c1, const double *__restrict__ c2, const double (&R)[9],
                                              double (&o)[9]) {
    double M[9], t[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            double s = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) s += R[a * 3 + c] * c1[c * 3 + b];
            M[a * 3 + b] = s;
        }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            double s = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) s += M[a * 3 + c] * R[b * 3 + c];
            t[a * 3 + b] = s + c2[a * 3 + b];
        }
    inv3(t, o);
}

struct FdfArgs {
    float T[12];  // T(x) = applyState(base, x), float
    float B[12];  // base_transformation_
};

// Double-double accumulation (error-free TwoSum of every term into a (hi, lo) pair): the thirteen
// sums come out as the correctly rounded value of the EXACT sum of their terms (up to ~1e-26
// relative), whatever the order they were added in.  Why it matters here and nowhere else: PCL's
// BFGS stops at a gradient tolerance of 1e-2 on this objective, evaluated through a float-quantised
// transform, so a last-bit difference in f or the gradient can flip a line-search branch and move the
// stopping point by millimetres.  With order-independent sums the CPU oracle (sequential) and this
// kernel (strided lanes, wave and block trees) agree bit for bit, and so does every decision after.
__device__ __forceinline__ void dd_add(double &hi, double &lo, double x) {
    const double s = hi + x;
    const double bb = s - hi;
    lo += (hi - (s - bb)) + (x - bb);
    hi = s;
}

template <int C, int M>
__device__ __forceinline__ void dd_halve(double (&hi)[kGicpAcc], double (&lo)[kGicpAcc], unsigned lane) {
    constexpr int H = (C + 1) / 2;
    const bool up = (lane & (unsigned) M) != 0u;
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const double h_lo = hi[i], l_lo = lo[i];
        const double h_hi = (H + i < C) ? hi[H + i] : 0.0, l_hi = (H + i < C) ? lo[H + i] : 0.0;
        const double sh = up ? h_lo : h_hi, sl = up ? l_lo : l_hi;  // the half this lane gives away
        double kh = up ? h_hi : h_lo, kl = up ? l_hi : l_lo;        // the half it keeps
        const double rh = __shfl_xor(sh, M), rl = __shfl_xor(sl, M);
        dd_add(kh, kl, rh);
        kl += rl;
        hi[i] = kh;
        lo[i] = kl;
    }
    if constexpr (M > 1) dd_halve<H, M / 2>(hi, lo, lane);
}
__device__ __forceinline__ int dd_comp_of_lane(unsigned lane) {
    int c = kGicpAcc, base = 0, valid = kGicpAcc;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        const int h = (c + 1) / 2;
        if (lane & (unsigned) m) {
            base += h;
            valid -= h;
        } else {
            valid = valid < h ? valid : h;
        }
        c = h;
    }
    return valid >= 1 ? base : -1;
}

// a[0] = sum r^T M r, a[1..3] = sum M r, a[4..12] = sum p_base (M r)^T; count separately.
// partials: [block][kGicpAcc][2] = (hi, lo) pairs.  The workgroup's share of one evaluation
// (all threads of the workgroup call it: it ends with a barrier and the row's store).
// one matched pair's terms (p: the source point, q: its match, M: the pair's Mahalanobis matrix)
__device__ __forceinline__ void gicp_fdf_point(double (&hi)[kGicpAcc], double (&lo)[kGicpAcc], const FdfArgs &A,
                                               float px, float py, float pz, float qx, float qy, float qz,
                                               const double (&M)[9]) {
    const float ppx = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(A.T[0], px), __fmul_rn(A.T[1], py)), __fmul_rn(A.T[2], pz)), A.T[3]);
    const float ppy = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(A.T[4], px), __fmul_rn(A.T[5], py)), __fmul_rn(A.T[6], pz)), A.T[7]);
    const float ppz = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(A.T[8], px), __fmul_rn(A.T[9], py)), __fmul_rn(A.T[10], pz)), A.T[11]);
    const double res[3] = {(double) __fsub_rn(ppx, qx), (double) __fsub_rn(ppy, qy), (double) __fsub_rn(ppz, qz)};
    double temp[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) temp[r] = M[r * 3] * res[0] + M[r * 3 + 1] * res[1] + M[r * 3 + 2] * res[2];
    dd_add(hi[0], lo[0], res[0] * temp[0] + res[1] * temp[1] + res[2] * temp[2]);
    const float pbx = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(A.B[0], px), __fmul_rn(A.B[1], py)), __fmul_rn(A.B[2], pz)), A.B[3]);
    const float pby = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(A.B[4], px), __fmul_rn(A.B[5], py)), __fmul_rn(A.B[6], pz)), A.B[7]);
    const float pbz = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(A.B[8], px), __fmul_rn(A.B[9], py)), __fmul_rn(A.B[10], pz)), A.B[11]);
    const double pb[3] = {(double) pbx, (double) pby, (double) pbz};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        dd_add(hi[1 + r], lo[1 + r], temp[r]);
#pragma unroll
        for (int c = 0; c < 3; ++c) dd_add(hi[4 + r * 3 + c], lo[4 + r * 3 + c], pb[r] * temp[c]);
    }
}

}  // namespace wm
