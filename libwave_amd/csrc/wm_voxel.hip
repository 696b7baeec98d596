// wm_voxel.hip -- pcl::VoxelGrid<PointXYZ>::filter and pcl::transformPointCloud on
// device, as libwave's matchers use them before every align:
//   wave_matching/src/icp.cpp:81-90,106-113   (filter both clouds, per scale)
//   wave_matching/src/icp.cpp:84-86           (pre-transform the filtered ref)
//   wave_matching/src/gicp.cpp:39-40,49-50
// Semantics follow PCL 1.8 filters/impl/voxel_grid.hpp: leaf index
// ijk = floor(p * (1/leaf)) - floor(min * (1/leaf)) in float, linear index
// i + j*dx + k*dx*dy, one float centroid (sequential float sum, ascending point
// index) per occupied leaf, output in ascending leaf-index order; non-finite points
// are skipped; if dx*dy*dz overflows int32 the input is returned unfiltered.
// The sort by leaf index is rocPRIM's stable radix sort (a plain library sort);
// everything else is hand-written.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "wm_internal.hpp"
#include "wm_sort.hpp"

namespace wm {

__global__ void __launch_bounds__(kBlock)
    k_vg_index(const float4 *__restrict__ in, unsigned n, float inv, int mbx, int mby, int mbz,
               int dx, int dxy, unsigned invalid, unsigned *__restrict__ idx,
               unsigned *__restrict__ perm) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = in[i];
    unsigned key = invalid;  // one past the last leaf: invalid points sort to the end
    if (p.x == p.x) {
        const int i0 = (int) (floorf(__fmul_rn(p.x, inv)) - (float) mbx);
        const int i1 = (int) (floorf(__fmul_rn(p.y, inv)) - (float) mby);
        const int i2 = (int) (floorf(__fmul_rn(p.z, inv)) - (float) mbz);
        key = (unsigned) (i0 + i1 * dx + i2 * dxy);
    }
    idx[i] = key;
    perm[i] = i;
}

// head flag per sorted position (1 where a new leaf starts); invalid tail gets 0
__global__ void __launch_bounds__(kBlock)
    k_vg_flags(const unsigned *__restrict__ idx_sorted, unsigned n, unsigned invalid,
               unsigned *__restrict__ flags) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const unsigned k = idx_sorted[i];
    flags[i] = (k != invalid && (i == 0 || idx_sorted[i - 1] != k)) ? 1u : 0u;
}

// heads[slot] = first sorted position of leaf `slot`; heads[n_leaves] = one past the last valid
// point (seg[n] = n_leaves, the scan's total)
__global__ void __launch_bounds__(kBlock)
    k_vg_heads(const unsigned *__restrict__ idx_sorted, const unsigned *__restrict__ flags,
               const unsigned *__restrict__ seg, unsigned n, unsigned invalid,
               unsigned *__restrict__ heads) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    if (flags[i]) heads[seg[i]] = i;
    if (idx_sorted[i] != invalid && (i + 1 == n || idx_sorted[i + 1] == invalid)) heads[seg[n]] = i + 1;
}

// one lane per leaf (compacted: every lane of a wave has a leaf -- with one lane per POINT and
// only the leaf heads working, a wave of a coarse grid ran a few lanes out of 64): sequential
// float sum over the leaf's points in ascending point index (the order is part of the result:
// float addition does not associate).  kVgTrip points per trip: their permutation / point loads are
// issued together, the adds stay in order.  The grid is sized for n leaves; lanes beyond the
// scan's total (read from the device) leave at once, so no host round trip precedes the launch.
constexpr int kVgTrip = 8;
constexpr unsigned kVgWaveAvg = 24;  // more points per leaf on average: one WAVE per leaf (below)
__global__ void __launch_bounds__(kBlock)
    k_vg_centroid(const float4 *__restrict__ in, const unsigned *__restrict__ perm_sorted,
                  const unsigned *__restrict__ heads, const unsigned *__restrict__ n_leaves,
                  unsigned n_valid, float4 *__restrict__ out) {
    const unsigned slot = blockIdx.x * kBlock + threadIdx.x;
    const unsigned leaves = *n_leaves;
    if (slot >= leaves || n_valid > kVgWaveAvg * (unsigned long long) leaves) return;  // big leaves: the wave kernel
    const unsigned i = heads[slot], j = heads[slot + 1];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (unsigned t = i; t < j; t += kVgTrip) {
        float4 p[kVgTrip];
#pragma unroll
        for (int u = 0; u < kVgTrip; ++u) p[u] = in[perm_sorted[t + u < j ? t + u : j - 1]];
#pragma unroll
        for (int u = 0; u < kVgTrip; ++u)
            if (t + u < j) {
                sx = __fadd_rn(sx, p[u].x);
                sy = __fadd_rn(sy, p[u].y);
                sz = __fadd_rn(sz, p[u].z);
            }
    }
    const float cnt = (float) (j - i);
    out[slot] = make_float4(__fdiv_rn(sx, cnt), __fdiv_rn(sy, cnt), __fdiv_rn(sz, cnt),
                            __uint_as_float(slot));
}

// Coarse grids (tens to thousands of points per leaf): one wave per leaf.  The sum must still be
// formed one point after the other, but the 64 lanes fetch 64 points at a time (the gather is the
// slow part) into LDS and lane 0 adds them in order; with one LANE per leaf a wave waited for its
// largest leaf while each lane gathered alone (168 us for the 0.8 m grid of a 1M-point cloud).
__global__ void __launch_bounds__(kBlock)
    k_vg_centroid_wave(const float4 *__restrict__ in, const unsigned *__restrict__ perm_sorted,
                       const unsigned *__restrict__ heads, const unsigned *__restrict__ n_leaves,
                       unsigned n_valid, float4 *__restrict__ out) {
    const unsigned leaves = *n_leaves;
    if (n_valid <= kVgWaveAvg * (unsigned long long) leaves) return;  // small leaves: the lane kernel
    __shared__ float s_p[kBlock / 64][3][64];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned waves_total = gridDim.x * (kBlock / 64);
    for (unsigned slot = blockIdx.x * (kBlock / 64) + wave; slot < leaves; slot += waves_total) {
        const unsigned i = heads[slot], j = heads[slot + 1];
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (unsigned t = i; t < j; t += 64) {
            if (t + lane < j) {
                const float4 p = in[perm_sorted[t + lane]];
                s_p[wave][0][lane] = p.x;
                s_p[wave][1][lane] = p.y;
                s_p[wave][2][lane] = p.z;
            }
            // (a wave's LDS traffic is in order: no barrier between its own write and read)
            if (lane == 0) {
                const unsigned m = j - t < 64u ? j - t : 64u;
                for (unsigned u = 0; u < m; ++u) {
                    sx = __fadd_rn(sx, s_p[wave][0][u]);
                    sy = __fadd_rn(sy, s_p[wave][1][u]);
                    sz = __fadd_rn(sz, s_p[wave][2][u]);
                }
            }
        }
        if (lane == 0) {
            const float cnt = (float) (j - i);
            out[slot] = make_float4(__fdiv_rn(sx, cnt), __fdiv_rn(sy, cnt), __fdiv_rn(sz, cnt),
                                    __uint_as_float(slot));
        }
    }
}

__global__ void __launch_bounds__(kBlock)
    k_copy_valid(const float4 *__restrict__ in, unsigned n, float4 *__restrict__ out) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) {
        float4 p = in[i];
        p.w = __uint_as_float(i);
        out[i] = p;
    }
}

int voxel_downsample_dev(wm_ctx *ctx, const float4 *in, size_t n, float leaf, float4 *out,
                         size_t *n_out, const VgKnown *known) {
    *n_out = 0;
    if (n == 0) return WM_OK;
    if (!(leaf > 0)) return WM_ERR_ARG;
    Bbox bb;
    size_t valid = 0;
    if (known) {
        bb = known->bb;
        valid = known->valid;
    } else {
        WM_TRY(compute_bbox(ctx, in, n, &bb, &valid));
    }
    if (valid == 0) return WM_OK;
    const float inv = 1.0f / leaf;
    const int64_t ex = (int64_t) ((bb.hi[0] - bb.lo[0]) * inv) + 1;
    const int64_t ey = (int64_t) ((bb.hi[1] - bb.lo[1]) * inv) + 1;
    const int64_t ez = (int64_t) ((bb.hi[2] - bb.lo[2]) * inv) + 1;
    const unsigned blocks = (unsigned) ((n + kBlock - 1) / kBlock);
    if (ex * ey * ez > (int64_t) INT32_MAX) {
        // PCL: "Leaf size is too small for the input dataset" -> output = input
        hipLaunchKernelGGL(k_copy_valid, dim3(blocks), dim3(kBlock), 0, ctx->stream, in,
                           (unsigned) n, out);
        WM_HIP(ctx, hipGetLastError());
        *n_out = n;
        return WM_OK;
    }
    int mb[3], db[3];
    for (int d = 0; d < 3; ++d) {
        mb[d] = (int) floorf(bb.lo[d] * inv);
        db[d] = (int) floorf(bb.hi[d] * inv) - mb[d] + 1;
    }
    WM_HIP(ctx, ctx->vg_idx.reserve(n * 4));
    WM_HIP(ctx, ctx->vg_idx2.reserve(n * 4));
    WM_HIP(ctx, ctx->vg_perm.reserve((n + 1) * 4));  // after the sort: the leaves' head positions (+ end)
    WM_HIP(ctx, ctx->vg_perm2.reserve(n * 4));
    WM_HIP(ctx, ctx->vg_seg.reserve((n + 1) * 4));
    unsigned *idx = ctx->vg_idx.as<unsigned>(), *idx2 = ctx->vg_idx2.as<unsigned>();
    unsigned *perm = ctx->vg_perm.as<unsigned>(), *perm2 = ctx->vg_perm2.as<unsigned>();
    unsigned *seg = ctx->vg_seg.as<unsigned>();
    // leaf indices run 0 .. total-1; `total` itself marks dropped points, so the sort only
    // needs the bits of `total` (coarse leaves: 2-3 radix passes instead of 4)
    const unsigned invalid = (unsigned) ((int64_t) db[0] * db[1] * db[2]);
    int bits = 1;
    while (bits < 32 && (invalid >> bits) != 0u) ++bits;
    hipLaunchKernelGGL(k_vg_index, dim3(blocks), dim3(kBlock), 0, ctx->stream, in, (unsigned) n,
                       inv, mb[0], mb[1], mb[2], db[0], db[0] * db[1], invalid, idx, perm);
    size_t tmp_bytes = 0;
    WM_HIP(ctx, sort_pairs_low_bits(nullptr, tmp_bytes, idx, idx2, perm, perm2, n, bits, ctx->stream,
                                    (size_t) ctx->tune_radix_min, ctx->tune_sort));
    WM_HIP(ctx, ctx->vg_tmp.reserve(tmp_bytes));
    WM_HIP(ctx, sort_pairs_low_bits(ctx->vg_tmp.p, tmp_bytes, idx, idx2, perm, perm2, n, bits, ctx->stream,
                                    (size_t) ctx->tune_radix_min, ctx->tune_sort));
    // head flags -> exclusive scan -> output slot per leaf; total = number of leaves
    hipLaunchKernelGGL(k_vg_flags, dim3(blocks), dim3(kBlock), 0, ctx->stream, idx2, (unsigned) n,
                       invalid, idx /* reuse as flags */);
    WM_TRY(exclusive_scan(ctx, idx, n, seg));
    unsigned *heads = perm;  // the sort's input permutation is dead by now
    hipLaunchKernelGGL(k_vg_heads, dim3(blocks), dim3(kBlock), 0, ctx->stream, idx2, idx, seg,
                       (unsigned) n, invalid, heads);
    // two launches, one of which leaves at once: which one is decided on the device from the
    // average leaf size (the leaf count is not on the host yet, and fetching it first would cost
    // more than the idle launch)
    hipLaunchKernelGGL(k_vg_centroid, dim3(blocks), dim3(kBlock), 0, ctx->stream, in, perm2, heads,
                       seg + n, (unsigned) valid, out);
    hipLaunchKernelGGL(k_vg_centroid_wave, dim3(2048), dim3(kBlock), 0, ctx->stream, in, perm2, heads,
                       seg + n, (unsigned) valid, out);
    WM_HIP(ctx, hipGetLastError());
    unsigned *h_total = (unsigned *) pinned_scratch(ctx, 0);
    if (!h_total) return WM_ERR_HIP;
    WM_TRY(fast_fetch(ctx, h_total, seg + n, 4));
    const unsigned total = *h_total;
    *n_out = total;
    return WM_OK;
}

// pcl::transformPointCloud(in, out, Eigen::Affine3d): double arithmetic
// ((T00*x + T01*y) + T02*z) + T03, float store; non-finite points stay non-finite
struct Mat34d {
    double m[12];
};

__global__ void __launch_bounds__(kBlock)
    k_transform_d(const float4 *__restrict__ in, unsigned n, Mat34d T, float4 *__restrict__ out) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = in[i];
    const double x = p.x, y = p.y, z = p.z;
    float4 o;
    o.x = (float) __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(T.m[0], x), __dmul_rn(T.m[1], y)),
                                      __dmul_rn(T.m[2], z)), T.m[3]);
    o.y = (float) __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(T.m[4], x), __dmul_rn(T.m[5], y)),
                                      __dmul_rn(T.m[6], z)), T.m[7]);
    o.z = (float) __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(T.m[8], x), __dmul_rn(T.m[9], y)),
                                      __dmul_rn(T.m[10], z)), T.m[11]);
    o.w = p.w;
    out[i] = o;
}

int transform_cloud_dev(wm_ctx *ctx, const float4 *in, size_t n, const double T[16], float4 *out) {
    if (n == 0) return WM_OK;
    Mat34d m;
    for (int k = 0; k < 12; ++k) m.m[k] = T[k];
    hipLaunchKernelGGL(k_transform_d, dim3((unsigned) ((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       ctx->stream, in, (unsigned) n, m, out);
    WM_HIP(ctx, hipGetLastError());
    return WM_OK;
}

// float4 (device) -> caller layout (stride >= 12), host or device destination
__global__ void __launch_bounds__(kBlock)
    k_unpack(const float4 *__restrict__ in, unsigned n, size_t stride, unsigned char *out) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = in[i];
    float *o = reinterpret_cast<float *>(out + (size_t) i * stride);
    o[0] = p.x;
    o[1] = p.y;
    o[2] = p.z;
    if (stride >= 16) o[3] = 1.0f;  // pcl::PointXYZ's padding member (data[3] = 1)
}

static int export_cloud(wm_ctx *ctx, const float4 *dev, size_t n, void *out, size_t stride, int mem) {
    if (n == 0) return WM_OK;
    unsigned char *dst = static_cast<unsigned char *>(out);
    if (mem == WM_MEM_HOST) {
        WM_HIP(ctx, ctx->staging.reserve(n * stride));
        dst = ctx->staging.as<unsigned char>();
        if (stride > 16) WM_HIP(ctx, hipMemsetAsync(dst, 0, n * stride, ctx->stream));
    }
    hipLaunchKernelGGL(k_unpack, dim3((unsigned) ((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       ctx->stream, dev, (unsigned) n, stride, dst);
    WM_HIP(ctx, hipGetLastError());
    if (mem == WM_MEM_HOST) WM_TRY(copy_to_caller(ctx, out, dst, n * stride));
    WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return WM_OK;
}

}  // namespace wm

using namespace wm;

extern "C" {

int wm_voxel_downsample(wm_ctx *ctx, const void *pts, size_t n, size_t stride, int mem, float leaf,
                        void *out, size_t out_stride, int out_mem, size_t cap, size_t *n_out) {
    if (!ctx || !n_out || (n > 0 && (!pts || !out)) || stride < 12 || (stride & 3) ||
        out_stride < 12 || (out_stride & 3) || !(leaf > 0) || n > 0x7FFFFFF0u)
        return WM_ERR_ARG;
    *n_out = 0;
    if (n == 0) return WM_OK;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    WM_HIP(ctx, ctx->io_a.reserve(n * sizeof(float4)));
    WM_HIP(ctx, ctx->io_b.reserve(n * sizeof(float4)));
    WM_TRY(pack_cloud(ctx, pts, n, stride, mem, ctx->io_a.as<float4>()));
    size_t m = 0;
    WM_TRY(voxel_downsample_dev(ctx, ctx->io_a.as<float4>(), n, leaf, ctx->io_b.as<float4>(), &m));
    if (m > cap) return WM_ERR_ARG;
    *n_out = m;
    return export_cloud(ctx, ctx->io_b.as<float4>(), m, out, out_stride, out_mem);
}

int wm_transform_cloud(wm_ctx *ctx, const void *pts, size_t n, size_t stride, int mem,
                       const double T[16], void *out, size_t out_stride, int out_mem) {
    if (!ctx || !T || (n > 0 && (!pts || !out)) || stride < 12 || (stride & 3) || out_stride < 12 ||
        (out_stride & 3) || n > 0x7FFFFFF0u)
        return WM_ERR_ARG;
    if (n == 0) return WM_OK;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    WM_HIP(ctx, ctx->io_a.reserve(n * sizeof(float4)));
    WM_HIP(ctx, ctx->io_b.reserve(n * sizeof(float4)));
    WM_TRY(pack_cloud(ctx, pts, n, stride, mem, ctx->io_a.as<float4>()));
    WM_TRY(transform_cloud_dev(ctx, ctx->io_a.as<float4>(), n, T, ctx->io_b.as<float4>()));
    return export_cloud(ctx, ctx->io_b.as<float4>(), n, out, out_stride, out_mem);
}

}  // extern "C"
