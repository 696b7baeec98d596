// wm_batch.hip -- the voxel-filtered branches of ICPMatcher::match() for MANY queued pairs at once
// (wave_matching/src/icp.cpp:77-104 multiscale, :105-122 single scale), the part of
// wm_icp_batch_match that wm_small.hip's resident registrations cannot do alone:
//   * pcl::VoxelGrid<PointXYZ>::filter for every cloud of the batch in ONE pass of device-wide
//     kernels (icp.cpp:81-90,106-113): the clouds are concatenated, the sort key carries the cloud
//     number above the leaf index, so one stable radix sort orders all of them; semantics exactly as
//     wm_voxel.hip (leaf index floor(p / leaf) - floor(min / leaf) in float, sequential float
//     centroid in ascending point index, leaves in ascending index);
//   * pcl::transformPointCloud of each filtered ref by its pair's running transform (icp.cpp:84-86);
//   * the resident registrations of the filtered pairs (wm_small.hip), scale by scale, every pair
//     carrying its own stopping-criteria state, running transform and fail-fast status.
// A pair is registered one scale at a time together with all the others; pairs that failed stop
// (icp.cpp:96-98).  Pairs whose filtered target does not fit the resident kernel (more than 65 535
// points), or whose leaf grid overflows int32 (PCL then returns the cloud unfiltered), are handed to
// wm_icp_match one by one.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "wm_internal.hpp"

#include <float.h>
#include <math.h>
#include <string.h>

#include <new>
#include <vector>

namespace wm {

constexpr int kVbThreads = 1024;

struct VbCloud {  // one cloud of the batch (device table)
    const unsigned char *raw;  // caller-layout points in device memory
    unsigned n;                // points
    unsigned off;              // its place in the concatenated packed / output arrays
};
struct VbBox {  // per cloud, found once (the same cloud is filtered at every scale)
    float lo[3], hi[3];
    unsigned n_valid;
    unsigned pad;
};
struct VbLeaf {  // per cloud and leaf size
    float inv;
    int mb[3];
    int dx, dxy;
    unsigned invalid;  // number of leaves = the key of dropped points
    int overflow;      // dx * dy * dz does not fit int32: PCL returns the input unfiltered
};

// the cloud a position of the concatenated arrays belongs to (clouds are laid out back to back)
__device__ __forceinline__ unsigned vb_cloud_of(const VbCloud *__restrict__ cl, unsigned n_clouds, unsigned g) {
    unsigned lo = 0, hi = n_clouds;  // last cloud with off <= g
    while (hi - lo > 1u) {
        const unsigned mid = (lo + hi) >> 1;
        if (cl[mid].off <= g) lo = mid;
        else hi = mid;
    }
    return lo;
}

// caller layout -> float4 (w = index within the cloud), non-finite points -> NaN (as k_pack, wm_grid.hip)
__global__ void __launch_bounds__(kBlock)
    k_vb_pack(const VbCloud *__restrict__ cl, unsigned n_clouds, unsigned total, unsigned stride, float4 *__restrict__ out) {
    const unsigned g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= total) return;
    const unsigned c = vb_cloud_of(cl, n_clouds, g);
    const unsigned i = g - cl[c].off;
    const float *p = reinterpret_cast<const float *>(cl[c].raw + (size_t) i * stride);
    float x = p[0], y = p[1], z = p[2];
    if (!(isfinite(x) && isfinite(y) && isfinite(z))) x = y = z = __builtin_nanf("");
    out[g] = make_float4(x, y, z, __uint_as_float(i));
}

// one workgroup per cloud: bounding box of its finite points and their number
__global__ void __launch_bounds__(kVbThreads)
    k_vb_bbox(const VbCloud *__restrict__ cl, const float4 *__restrict__ pts, VbBox *__restrict__ box) {
    const VbCloud c = cl[blockIdx.x];
    float lo0 = INFINITY, lo1 = INFINITY, lo2 = INFINITY, hi0 = -INFINITY, hi1 = -INFINITY, hi2 = -INFINITY;
    unsigned cnt = 0;
    for (unsigned i = threadIdx.x; i < c.n; i += kVbThreads) {
        const float4 p = pts[c.off + i];
        if (p.x == p.x) {
            lo0 = fminf(lo0, p.x), lo1 = fminf(lo1, p.y), lo2 = fminf(lo2, p.z);
            hi0 = fmaxf(hi0, p.x), hi1 = fmaxf(hi1, p.y), hi2 = fmaxf(hi2, p.z);
            ++cnt;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo0 = fminf(lo0, __shfl_xor(lo0, off)), lo1 = fminf(lo1, __shfl_xor(lo1, off));
        lo2 = fminf(lo2, __shfl_xor(lo2, off)), hi0 = fmaxf(hi0, __shfl_xor(hi0, off));
        hi1 = fmaxf(hi1, __shfl_xor(hi1, off)), hi2 = fmaxf(hi2, __shfl_xor(hi2, off));
        cnt += __shfl_xor(cnt, off);
    }
    __shared__ float s_b[kVbThreads / 64][6];
    __shared__ unsigned s_c[kVbThreads / 64];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s_b[wave][0] = lo0, s_b[wave][1] = lo1, s_b[wave][2] = lo2;
        s_b[wave][3] = hi0, s_b[wave][4] = hi1, s_b[wave][5] = hi2;
        s_c[wave] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kVbThreads / 64; ++w) {
            lo0 = fminf(lo0, s_b[w][0]), lo1 = fminf(lo1, s_b[w][1]), lo2 = fminf(lo2, s_b[w][2]);
            hi0 = fmaxf(hi0, s_b[w][3]), hi1 = fmaxf(hi1, s_b[w][4]), hi2 = fmaxf(hi2, s_b[w][5]);
            cnt += s_c[w];
        }
        VbBox b;
        b.lo[0] = lo0, b.lo[1] = lo1, b.lo[2] = lo2, b.hi[0] = hi0, b.hi[1] = hi1, b.hi[2] = hi2;
        b.n_valid = cnt;
        b.pad = 0;
        box[blockIdx.x] = b;
    }
}

// per cloud: the leaf lattice of this scale (the host arithmetic of voxel_downsample_dev, wm_voxel.hip)
__global__ void __launch_bounds__(kBlock)
    k_vb_leaf(const VbBox *__restrict__ box, unsigned n_clouds, float leaf, VbLeaf *__restrict__ out) {
    const unsigned c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= n_clouds) return;
    const VbBox b = box[c];
    VbLeaf l;
    l.inv = __fdiv_rn(1.0f, leaf);
    l.overflow = 0;
    l.mb[0] = l.mb[1] = l.mb[2] = 0;
    l.dx = l.dxy = 1;
    l.invalid = 0;
    if (b.n_valid > 0) {
        const long long ex = (long long) __fmul_rn(__fsub_rn(b.hi[0], b.lo[0]), l.inv) + 1;
        const long long ey = (long long) __fmul_rn(__fsub_rn(b.hi[1], b.lo[1]), l.inv) + 1;
        const long long ez = (long long) __fmul_rn(__fsub_rn(b.hi[2], b.lo[2]), l.inv) + 1;
        if (ex * ey * ez > 2147483647ll) {
            l.overflow = 1;
        } else {
            int db[3];
            for (int d = 0; d < 3; ++d) {
                l.mb[d] = (int) floorf(__fmul_rn(b.lo[d], l.inv));
                db[d] = (int) floorf(__fmul_rn(b.hi[d], l.inv)) - l.mb[d] + 1;
            }
            l.dx = db[0];
            l.dxy = db[0] * db[1];
            l.invalid = (unsigned) ((long long) db[0] * db[1] * db[2]);
        }
    }
    out[c] = l;
}

// key = cloud << shift | leaf index (k_vg_index's); dropped points and the clouds that are not filtered
// at this scale get the cloud's `invalid`, which sorts behind its leaves
__global__ void __launch_bounds__(kBlock)
    k_vb_index(const VbCloud *__restrict__ cl, unsigned n_clouds, unsigned total, const float4 *__restrict__ pts,
               const VbLeaf *__restrict__ leaf, const unsigned char *__restrict__ skip, unsigned shift,
               unsigned long long *__restrict__ key, unsigned *__restrict__ perm) {
    const unsigned g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= total) return;
    const unsigned c = vb_cloud_of(cl, n_clouds, g);
    const VbLeaf l = leaf[c];
    const float4 p = pts[g];
    unsigned k = l.invalid;
    if (p.x == p.x && !l.overflow && !skip[c]) {
        const int i0 = (int) (floorf(__fmul_rn(p.x, l.inv)) - (float) l.mb[0]);
        const int i1 = (int) (floorf(__fmul_rn(p.y, l.inv)) - (float) l.mb[1]);
        const int i2 = (int) (floorf(__fmul_rn(p.z, l.inv)) - (float) l.mb[2]);
        k = (unsigned) (i0 + i1 * l.dx + i2 * l.dxy);
    }
    key[g] = ((unsigned long long) c << shift) | k;  // (shift = bits of the batch's largest leaf count: fewer radix passes)
    perm[g] = g;
}

// 1 where a new group of equal keys starts (leaves AND the dropped tail of every cloud: the tail is a
// group of its own that the centroid pass skips)
__global__ void __launch_bounds__(kBlock)
    k_vb_flags(const unsigned long long *__restrict__ key_sorted, unsigned total, unsigned *__restrict__ flags) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= total) return;
    flags[i] = (i == 0 || key_sorted[i - 1] != key_sorted[i]) ? 1u : 0u;
}

__global__ void __launch_bounds__(kBlock)
    k_vb_heads(const unsigned *__restrict__ flags, const unsigned *__restrict__ seg, unsigned total,
               unsigned *__restrict__ heads) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= total) return;
    if (flags[i]) heads[seg[i]] = i;
    if (i + 1 == total) heads[seg[total]] = total;
}

// one lane per group: sequential float sum over the leaf's points in ascending point index (the order
// is part of the result), kVbTrip loads in flight; output slot = the leaf's rank within its cloud
constexpr int kVbTrip = 8;
__global__ void __launch_bounds__(kBlock)
    k_vb_centroid(const VbCloud *__restrict__ cl, const VbLeaf *__restrict__ leaf, const float4 *__restrict__ pts,
                  const unsigned long long *__restrict__ key_sorted, const unsigned *__restrict__ perm_sorted,
                  const unsigned *__restrict__ heads, const unsigned *__restrict__ seg, unsigned total, unsigned shift,
                  float4 *__restrict__ out) {
    const unsigned slot = blockIdx.x * kBlock + threadIdx.x;
    if (slot >= seg[total]) return;
    const unsigned i = heads[slot], j = heads[slot + 1];
    const unsigned long long key = key_sorted[i];
    const unsigned c = (unsigned) (key >> shift);
    if ((unsigned) (key & ((1ull << shift) - 1ull)) == leaf[c].invalid) return;  // the cloud's dropped points
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (unsigned t = i; t < j; t += kVbTrip) {
        float4 p[kVbTrip];
#pragma unroll
        for (int u = 0; u < kVbTrip; ++u) p[u] = pts[perm_sorted[t + u < j ? t + u : j - 1]];
#pragma unroll
        for (int u = 0; u < kVbTrip; ++u)
            if (t + u < j) {
                sx = __fadd_rn(sx, p[u].x);
                sy = __fadd_rn(sy, p[u].y);
                sz = __fadd_rn(sz, p[u].z);
            }
    }
    const float cnt = (float) (j - i);
    const unsigned rank = slot - seg[cl[c].off];
    out[cl[c].off + rank] = make_float4(__fdiv_rn(sx, cnt), __fdiv_rn(sy, cnt), __fdiv_rn(sz, cnt), __uint_as_float(rank));
}

// leaves per cloud: the groups between its first sorted position and its first dropped point
__global__ void __launch_bounds__(kBlock)
    k_vb_counts(const VbCloud *__restrict__ cl, const VbBox *__restrict__ box, const VbLeaf *__restrict__ leaf,
                const unsigned char *__restrict__ skip, const unsigned *__restrict__ seg, unsigned n_clouds,
                unsigned *__restrict__ n_out) {
    const unsigned c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= n_clouds) return;
    unsigned n = 0;
    if (leaf[c].overflow) n = 0xFFFFFFFFu;  // (the host hands this pair to the one-pair path)
    else if (!skip[c]) n = seg[cl[c].off + box[c].n_valid] - seg[cl[c].off];
    n_out[c] = n;
}

// pcl::transformPointCloud(cloud, cloud, Eigen::Affine3d), in place, every cloud by its own matrix
// (k_transform_d, wm_voxel.hip); one workgroup column per cloud
struct VbXform {
    double m[12];
    unsigned off, n;
    unsigned on, pad;
};
__global__ void __launch_bounds__(kBlock)
    k_vb_transform(const VbXform *__restrict__ xf, float4 *__restrict__ pts) {
    const VbXform X = xf[blockIdx.y];
    if (!X.on) return;
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < X.n; i += gridDim.x * kBlock) {
        const float4 p = pts[X.off + i];
        const double x = p.x, y = p.y, z = p.z;
        float4 o;
        o.x = (float) __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(X.m[0], x), __dmul_rn(X.m[1], y)), __dmul_rn(X.m[2], z)), X.m[3]);
        o.y = (float) __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(X.m[4], x), __dmul_rn(X.m[5], y)), __dmul_rn(X.m[6], z)), X.m[7]);
        o.z = (float) __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(X.m[8], x), __dmul_rn(X.m[9], y)), __dmul_rn(X.m[10], z)), X.m[11]);
        o.w = p.w;
        pts[X.off + i] = o;
    }
}

// ------------------------------------------------------------------ host side
struct BatchVoxel {
    DevBuf raw;        // the uploaded clouds (host input)
    DevBuf table;      // VbCloud[], VbBox[], VbLeaf[], skip[], n_out[], VbXform[]
    DevBuf packed, out;
    DevBuf key, key2, perm, perm2, flags, seg, heads, tmp;
    void *h_raw = nullptr;
    size_t h_raw_cap = 0;
    void *h_tab = nullptr;
    size_t h_tab_cap = 0;
};

static BatchVoxel *voxel_of(wm_ctx *ctx) {
    if (!ctx->batch_voxel) ctx->batch_voxel = new (std::nothrow) BatchVoxel();
    return static_cast<BatchVoxel *>(ctx->batch_voxel);
}

void batch_voxel_release(wm_ctx *ctx) {
    BatchVoxel *b = static_cast<BatchVoxel *>(ctx->batch_voxel);
    if (!b) return;
    DevBuf *bufs[] = {&b->raw, &b->table, &b->packed, &b->out, &b->key, &b->key2, &b->perm, &b->perm2, &b->flags, &b->seg, &b->heads, &b->tmp};
    for (DevBuf *d : bufs) d->release();
    if (b->h_raw) (void) hipHostFree(b->h_raw);
    if (b->h_tab) (void) hipHostFree(b->h_tab);
    delete b;
    ctx->batch_voxel = nullptr;
}

static int pinned_grow(wm_ctx *ctx, void **p, size_t *cap, size_t bytes) {
    if (bytes <= *cap) return WM_OK;
    if (*p) (void) hipHostFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    WM_HIP(ctx, hipHostMalloc(p, want, hipHostMallocDefault));
    *cap = want;
    return WM_OK;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// The clouds of one sub-batch on the device, packed, with their bounding boxes: what every scale's
// filter starts from.  Cloud 2 j = ref of pair j, 2 j + 1 = its target.
struct VoxelBatch {
    wm_ctx *ctx = nullptr;
    BatchVoxel *V = nullptr;
    unsigned n_pairs = 0, n_clouds = 0, cloud_bits = 1;
    std::vector<VbBox> h_box;  // the clouds' bounding boxes on the host: they size the sort key of every scale
    size_t total = 0;
    std::vector<VbCloud> cl;
    std::vector<unsigned> n_out;
    size_t o_cloud = 0, o_box = 0, o_leaf = 0, o_skip = 0, o_nout = 0, o_xf = 0, tab_bytes = 0;
    unsigned char *ht = nullptr, *dt = nullptr;
    unsigned char *h_skip = nullptr;
    float4 *packed = nullptr, *filtered = nullptr;

    int setup(wm_ctx *c, const wm_batch_item *items, const std::vector<int> &idx, size_t stride, int mem) {
        ctx = c;
        V = voxel_of(ctx);
        if (!V) return WM_ERR_NOMEM;
        n_pairs = (unsigned) idx.size();
        n_clouds = 2u * n_pairs;
        cl.assign(n_clouds, VbCloud{nullptr, 0, 0});
        total = 0;
        size_t raw_bytes = 0;
        for (unsigned j = 0; j < n_pairs; ++j) {
            const wm_batch_item &it = items[idx[j]];
            const size_t n2[2] = {it.n_src, it.n_target};
            for (int s = 0; s < 2; ++s) {
                cl[2 * j + s].n = (unsigned) n2[s];
                cl[2 * j + s].off = (unsigned) total;
                total += n2[s];
                raw_bytes += align_up(n2[s] * stride, 16);
            }
        }
        if (total == 0 || total > 0x7FFFFFF0u) return WM_ERR_ARG;
        o_cloud = 0;
        o_box = align_up(o_cloud + n_clouds * sizeof(VbCloud), 256);
        o_leaf = align_up(o_box + n_clouds * sizeof(VbBox), 256);
        o_skip = align_up(o_leaf + n_clouds * sizeof(VbLeaf), 256);
        o_nout = align_up(o_skip + n_clouds, 256);
        o_xf = align_up(o_nout + n_clouds * sizeof(unsigned), 256);
        tab_bytes = align_up(o_xf + n_pairs * sizeof(VbXform), 256);
        WM_HIP(ctx, V->table.reserve(tab_bytes));
        WM_HIP(ctx, V->packed.reserve(total * sizeof(float4)));
        WM_HIP(ctx, V->out.reserve(total * sizeof(float4)));
        WM_HIP(ctx, V->key.reserve(total * 8));
        WM_HIP(ctx, V->key2.reserve(total * 8));
        WM_HIP(ctx, V->perm.reserve(total * 4));
        WM_HIP(ctx, V->perm2.reserve(total * 4));
        WM_HIP(ctx, V->flags.reserve(total * 4));
        WM_HIP(ctx, V->seg.reserve((total + 1) * 4));
        WM_HIP(ctx, V->heads.reserve((total + 2) * 4));
        WM_TRY(pinned_grow(ctx, &V->h_tab, &V->h_tab_cap, tab_bytes));
        WM_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (the staging buffers may still feed the previous sub-batch)
        ht = static_cast<unsigned char *>(V->h_tab);
        dt = V->table.as<unsigned char>();
        // the clouds onto the device (host input: through pinned memory, DMA of one slice under the next copy)
        if (mem == WM_MEM_HOST) {
            WM_HIP(ctx, V->raw.reserve(raw_bytes));
            WM_TRY(pinned_grow(ctx, &V->h_raw, &V->h_raw_cap, raw_bytes));
            unsigned char *h = static_cast<unsigned char *>(V->h_raw), *d = V->raw.as<unsigned char>();
            size_t off = 0, sent = 0;
            for (unsigned j = 0; j < n_pairs; ++j) {
                const wm_batch_item &it = items[idx[j]];
                const void *src2[2] = {it.src, it.target};
                const size_t n2[2] = {it.n_src, it.n_target};
                for (int s = 0; s < 2; ++s) {
                    if (n2[s]) memcpy(h + off, src2[s], n2[s] * stride);
                    cl[2 * j + s].raw = d + off;
                    off += align_up(n2[s] * stride, 16);
                    if (off - sent >= ((size_t) 2 << 20)) {
                        WM_HIP(ctx, hipMemcpyAsync(d + sent, h + sent, off - sent, hipMemcpyHostToDevice, ctx->stream));
                        sent = off;
                    }
                }
            }
            if (off > sent) WM_HIP(ctx, hipMemcpyAsync(d + sent, h + sent, off - sent, hipMemcpyHostToDevice, ctx->stream));
        } else {
            for (unsigned j = 0; j < n_pairs; ++j) {
                cl[2 * j].raw = static_cast<const unsigned char *>(items[idx[j]].src);
                cl[2 * j + 1].raw = static_cast<const unsigned char *>(items[idx[j]].target);
            }
        }
        memcpy(ht + o_cloud, cl.data(), n_clouds * sizeof(VbCloud));
        h_skip = ht + o_skip;
        memset(h_skip, 0, n_clouds);
        WM_HIP(ctx, hipMemcpyAsync(dt, ht, tab_bytes, hipMemcpyHostToDevice, ctx->stream));
        packed = V->packed.as<float4>();
        filtered = V->out.as<float4>();
        const unsigned blocks = (unsigned) ((total + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(k_vb_pack, dim3(blocks), dim3(kBlock), 0, ctx->stream, d_cl(), n_clouds, (unsigned) total, (unsigned) stride, packed);
        hipLaunchKernelGGL(k_vb_bbox, dim3(n_clouds), dim3(kVbThreads), 0, ctx->stream, d_cl(), packed, d_box());
        WM_HIP(ctx, hipGetLastError());
        cloud_bits = 1;
        while ((1u << cloud_bits) < n_clouds) ++cloud_bits;
        h_box.resize(n_clouds);
        WM_HIP(ctx, hipMemcpyAsync(h_box.data(), d_box(), n_clouds * sizeof(VbBox), hipMemcpyDeviceToHost, ctx->stream));
        WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        n_out.assign(n_clouds, 0);
        return WM_OK;
    }
    const VbCloud *d_cl() const { return reinterpret_cast<const VbCloud *>(dt + o_cloud); }
    VbBox *d_box() const { return reinterpret_cast<VbBox *>(dt + o_box); }
    VbLeaf *d_leaf() const { return reinterpret_cast<VbLeaf *>(dt + o_leaf); }
    unsigned char *d_skip() const { return dt + o_skip; }
    unsigned *d_nout() const { return reinterpret_cast<unsigned *>(dt + o_nout); }
    VbXform *d_xf() const { return reinterpret_cast<VbXform *>(dt + o_xf); }

    // pcl::VoxelGrid of every cloud whose h_skip entry is 0: centroids into `filtered` at the cloud's
    // offset, their number into n_out (0xFFFFFFFF: the leaf lattice overflows int32)
    int filter(float leaf) {
        const unsigned blocks = (unsigned) ((total + kBlock - 1) / kBlock), cblocks = (n_clouds + kBlock - 1) / kBlock;
        // the widest leaf count of the batch at this scale (k_vb_leaf's arithmetic) -> bits of the key's leaf part
        unsigned shift = 1;
        {
            const float inv = 1.0f / leaf;
            unsigned long long most = 1;
            for (unsigned c = 0; c < n_clouds; ++c) {
                const VbBox &b = h_box[c];
                if (!b.n_valid) continue;
                unsigned long long cells = 1;
                for (int d = 0; d < 3; ++d) {
                    const long long m = (long long) floorf(b.lo[d] * inv), M = (long long) floorf(b.hi[d] * inv);
                    cells *= (unsigned long long) (M - m + 1);
                }
                if (cells <= 2147483647ull && cells > most) most = cells;  // (beyond int32: overflow, key = 0)
            }
            while (shift < 32 && (most >> shift) != 0ull) ++shift;
        }
        const unsigned bits = shift + cloud_bits;
        WM_HIP(ctx, hipMemcpyAsync(d_skip(), h_skip, n_clouds, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_vb_leaf, dim3(cblocks), dim3(kBlock), 0, ctx->stream, d_box(), n_clouds, leaf, d_leaf());
        unsigned long long *key = V->key.as<unsigned long long>(), *key2 = V->key2.as<unsigned long long>();
        unsigned *perm = V->perm.as<unsigned>(), *perm2 = V->perm2.as<unsigned>();
        unsigned *flags = V->flags.as<unsigned>(), *seg = V->seg.as<unsigned>(), *heads = V->heads.as<unsigned>();
        hipLaunchKernelGGL(k_vb_index, dim3(blocks), dim3(kBlock), 0, ctx->stream, d_cl(), n_clouds, (unsigned) total, packed, d_leaf(),
                           d_skip(), shift, key, perm);
        WM_HIP(ctx, hipGetLastError());
        size_t tmp_bytes = 0;
        WM_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, key, key2, perm, perm2, total, 0u, bits, ctx->stream));
        WM_HIP(ctx, V->tmp.reserve(tmp_bytes));
        WM_HIP(ctx, rocprim::radix_sort_pairs(V->tmp.p, tmp_bytes, key, key2, perm, perm2, total, 0u, bits, ctx->stream));
        hipLaunchKernelGGL(k_vb_flags, dim3(blocks), dim3(kBlock), 0, ctx->stream, key2, (unsigned) total, flags);
        WM_TRY(exclusive_scan(ctx, flags, total, seg));
        hipLaunchKernelGGL(k_vb_heads, dim3(blocks), dim3(kBlock), 0, ctx->stream, flags, seg, (unsigned) total, heads);
        hipLaunchKernelGGL(k_vb_centroid, dim3(blocks), dim3(kBlock), 0, ctx->stream, d_cl(), d_leaf(), packed, key2, perm2, heads, seg,
                           (unsigned) total, shift, filtered);
        hipLaunchKernelGGL(k_vb_counts, dim3(cblocks), dim3(kBlock), 0, ctx->stream, d_cl(), d_box(), d_leaf(), d_skip(), seg, n_clouds,
                           d_nout());
        WM_HIP(ctx, hipGetLastError());
        WM_HIP(ctx, hipMemcpyAsync(n_out.data(), d_nout(), n_clouds * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return WM_OK;
    }
};

// One sub-batch (bounded size) of voxel-filtered matches.  `idx` = the items it holds.
static int scaled_sub_batch(wm_ctx *ctx, const wm_batch_item *items, const std::vector<int> &idx, size_t stride, int mem,
                            const wm_icp_params *p, float res, int steps, int with_info, double *T_out, double *info_out,
                            wm_icp_stats *stats, int *status, std::vector<int> &one_by_one) {
    // a sub-batch of empty pairs only (PCL: empty input -> "Not enough correspondences"): nothing to
    // filter, every item gets its status, the batch has run
    {
        size_t total_pts = 0;
        for (int k : idx) total_pts += items[k].n_src + items[k].n_target;
        if (total_pts == 0) {
            for (int k : idx) {
                status[k] = WM_ERR_STATE;
                if (stats) stats[k].state = WM_CONV_NO_CORRESPONDENCES;
            }
            return WM_OK;
        }
    }
    VoxelBatch vb;
    WM_TRY(vb.setup(ctx, items, idx, stride, mem));
    const unsigned n_pairs = vb.n_pairs;
    const std::vector<VbCloud> &cl = vb.cl;
    std::vector<unsigned> &n_out = vb.n_out;
    float4 *filtered = vb.filtered;
    // ---- per pair state across the scales
    struct PairState {
        double running[16];
        double prev_mse;
        bool alive;
    };
    std::vector<PairState> ps(n_pairs);
    for (unsigned j = 0; j < n_pairs; ++j) {
        mat4_identity(ps[j].running);
        ps[j].prev_mse = DBL_MAX;  // every pair starts with fresh stopping criteria
        ps[j].alive = true;
        const wm_batch_item &it = items[idx[j]];
        if (it.n_src == 0 || it.n_target == 0) {  // PCL: empty input -> "Not enough correspondences"
            status[idx[j]] = it.n_src == 0 && it.n_target == 0 ? WM_ERR_STATE : WM_TOO_FEW_CORRESPONDENCES;
            if (stats) stats[idx[j]].state = WM_CONV_NO_CORRESPONDENCES;
            ps[j].alive = false;
        } else {
            status[idx[j]] = WM_OK;
        }
    }
    std::vector<SmallJob> jobs;
    std::vector<unsigned> job_pair;
    std::vector<SmallResult> got;
    wm_icp_params prm = *p;
    for (int i = steps; i >= 0; --i) {
        unsigned alive = 0;
        for (unsigned j = 0; j < n_pairs; ++j) {
            vb.h_skip[2 * j] = vb.h_skip[2 * j + 1] = ps[j].alive ? 0 : 1;
            alive += ps[j].alive;
        }
        if (!alive) break;
        const float leaf = (float) (pow(2, i) * res);  // icp.cpp:80
        WM_TRY(vb.filter(leaf));
        // pairs this path cannot take any further: the one-pair path registers them from scratch
        for (unsigned j = 0; j < n_pairs; ++j) {
            if (!ps[j].alive) continue;
            if (n_out[2 * j] == 0xFFFFFFFFu || n_out[2 * j + 1] == 0xFFFFFFFFu || n_out[2 * j + 1] > (unsigned) WM_BATCH_MAX_TARGET_POINTS) {
                ps[j].alive = false;
                one_by_one.push_back(idx[j]);
            }
        }
        if (steps > 0) {
            // icp.cpp:84-86: the filtered ref goes through the transform found so far; icp.cpp:93-94
            VbXform *hx = reinterpret_cast<VbXform *>(vb.ht + vb.o_xf);
            unsigned longest = 0;
            for (unsigned j = 0; j < n_pairs; ++j) {
                for (int k = 0; k < 12; ++k) hx[j].m[k] = ps[j].running[k];
                hx[j].off = cl[2 * j].off;
                hx[j].n = ps[j].alive ? n_out[2 * j] : 0;
                hx[j].on = ps[j].alive ? 1 : 0;
                hx[j].pad = 0;
                longest = hx[j].n > longest ? hx[j].n : longest;
            }
            if (longest) {
                WM_HIP(ctx, hipMemcpyAsync(vb.d_xf(), hx, n_pairs * sizeof(VbXform), hipMemcpyHostToDevice, ctx->stream));
                hipLaunchKernelGGL(k_vb_transform, dim3((longest + kBlock - 1) / kBlock, n_pairs), dim3(kBlock), 0, ctx->stream, vb.d_xf(), filtered);
                WM_HIP(ctx, hipGetLastError());
            }
            prm.max_corr = pow(2, i) * p->max_corr;
        }
        // ---- the resident registrations of this scale
        jobs.clear();
        job_pair.clear();
        for (unsigned j = 0; j < n_pairs; ++j) {
            if (!ps[j].alive) continue;
            const unsigned nr = n_out[2 * j], nt = n_out[2 * j + 1];
            if (nr == 0 || nt == 0) {  // nothing left of a cloud: the align fails as on empty input
                status[idx[j]] = WM_TOO_FEW_CORRESPONDENCES;
                if (stats) stats[idx[j]].state = WM_CONV_NO_CORRESPONDENCES;
                ps[j].alive = false;
                continue;
            }
            jobs.push_back(SmallJob{filtered + cl[2 * j].off, nr, filtered + cl[2 * j + 1].off, nt, ps[j].prev_mse, 1});
            job_pair.push_back(j);
        }
        if (jobs.empty()) continue;
        got.resize(jobs.size());
        float ms = 0;
        // estimateInfo() works on the clouds of the LAST align, whichever scale that turns out to be
        WM_TRY(small_run(ctx, jobs.data(), (int) jobs.size(), sizeof(float4), WM_MEM_DEVICE, &prm, with_info, p->max_corr, got.data(), &ms));
        for (size_t q = 0; q < jobs.size(); ++q) {
            const unsigned j = job_pair[q];
            const int k = idx[j];
            const SmallResult &r = got[q];
            ps[j].prev_mse = r.prev_mse;  // (one PCL object per matcher: its criteria keep the last MSE across the scales)
            if (stats) {
                const float before = stats[k].align_ms;
                small_fill_stats(r, ms, &stats[k]);
                stats[k].align_ms += before;
            }
            if (with_info && info_out) memcpy(info_out + 36 * (size_t) k, r.info, sizeof(r.info));
            if (r.state == WM_CONV_NO_CORRESPONDENCES || !r.converged) {  // icp.cpp:96-98: fail fast, result untouched
                status[k] = r.state == WM_CONV_NO_CORRESPONDENCES ? WM_TOO_FEW_CORRESPONDENCES : WM_NOT_CONVERGED;
                ps[j].alive = false;
                continue;
            }
            mat4_mul(r.T, ps[j].running, ps[j].running);  // icp.cpp:99-101
        }
    }
    for (unsigned j = 0; j < n_pairs; ++j)
        if (ps[j].alive && status[idx[j]] == WM_OK && T_out) memcpy(T_out + 16 * (size_t) idx[j], ps[j].running, sizeof(ps[j].running));
    return WM_OK;
}

// pcl::VoxelGrid of every cloud of `idx`'s items (cloud 2 j = the ref of item idx[j], 2 j + 1 = its target) in one
// pass: *filtered + off[c] = cloud c's centroids (float4, device memory of this context, valid until the next batched
// filter), n_out[c] their number (0xFFFFFFFF: the leaf lattice overflows int32 -- PCL returns that cloud unfiltered).
int batch_voxel_filter(wm_ctx *ctx, const wm_batch_item *items, const std::vector<int> &idx, size_t stride, int mem, float leaf,
                       const float4 **filtered, std::vector<unsigned> &off, std::vector<unsigned> &n_out) {
    off.assign(2 * idx.size(), 0u);
    n_out.assign(2 * idx.size(), 0u);
    *filtered = nullptr;
    size_t total_pts = 0;
    for (int k : idx) total_pts += items[k].n_src + items[k].n_target;
    if (total_pts == 0) return WM_OK;
    VoxelBatch vb;
    WM_TRY(vb.setup(ctx, items, idx, stride, mem));
    WM_TRY(vb.filter(leaf));
    *filtered = vb.filtered;
    for (unsigned c = 0; c < vb.n_clouds; ++c) {
        off[c] = vb.cl[c].off;
        n_out[c] = vb.n_out[c];
    }
    return WM_OK;
}

int batch_match_scaled(wm_ctx *ctx, const wm_batch_item *items, int n_items, size_t stride, int mem,
                       const wm_icp_params *p, float res, int multiscale_steps, int with_info, double *T_out,
                       double *info_out, wm_icp_stats *stats, int *status) {
    const int steps = multiscale_steps > 0 ? multiscale_steps : 0;
    // sub-batches of bounded size: every cloud costs ~70 bytes per point of sort and staging buffers
    // (a sub-batch should hold enough pairs to fill the device's 256 compute units with resident
    // registrations: 16 M points = 145 pairs of two 55 000-point scans, ~1.1 GB of buffers)
    const size_t budget = (size_t) 16 << 20;  // points per sub-batch
    std::vector<int> idx, one_by_one;
    size_t pts = 0;
    for (int k = 0; k <= n_items; ++k) {
        const size_t need = k < n_items ? items[k].n_src + items[k].n_target : 0;
        if (k == n_items || (!idx.empty() && pts + need > budget)) {
            if (!idx.empty())
                WM_TRY(scaled_sub_batch(ctx, items, idx, stride, mem, p, res, steps, with_info, T_out, info_out, stats, status, one_by_one));
            idx.clear();
            pts = 0;
        }
        if (k < n_items) {
            idx.push_back(k);
            pts += need;
        }
    }
    for (int k : one_by_one) {  // (leaf lattice beyond int32, or a filtered target beyond the resident kernel)
        // a pair of its own starts with fresh stopping criteria, but its SCALES carry the last MSE from one
        // align to the next, as the batched path and the reference's one PCL object do
        wm_icp_params q = *p;
        q.carry_state = 1;
        ctx->prev_mse = -1;
        double T[16];
        wm_icp_stats s;
        const int rc = wm_icp_match(ctx, items[k].src, items[k].n_src, items[k].target, items[k].n_target, stride, mem, &q, res,
                                    multiscale_steps, T, &s);
        if (rc < 0) return rc;
        status[k] = rc;
        if (stats) stats[k] = s;
        if (rc == WM_OK && T_out) memcpy(T_out + 16 * (size_t) k, T, sizeof(T));
        if (with_info && info_out) {
            double info[36];
            int deg = 0;
            if (wm_icp_info(ctx, WM_INFO_LUMOLD, nullptr, 0, 0, p->max_corr, info, &deg) == WM_OK)
                memcpy(info_out + 36 * (size_t) k, info, sizeof(info));
        }
    }
    return WM_OK;
}

}  // namespace wm

using namespace wm;

extern "C" {

int wm_voxel_downsample_batch(wm_ctx *ctx, const wm_batch_item *items, int n_items, size_t stride, int mem, float leaf,
                              float *out_xyz, size_t cap_points, size_t *n_out) {
    if (!ctx || n_items < 0 || (n_items > 0 && !items) || !n_out || (cap_points > 0 && !out_xyz) || stride < 12 || (stride & 3) ||
        !(leaf > 0))
        return WM_ERR_ARG;
    for (int k = 0; k < 2 * n_items; ++k) n_out[k] = 0;
    if (n_items == 0) return WM_OK;
    size_t pts = 0;
    std::vector<int> idx((size_t) n_items);
    for (int k = 0; k < n_items; ++k) {
        idx[(size_t) k] = k;
        if ((items[k].n_src > 0 && !items[k].src) || (items[k].n_target > 0 && !items[k].target)) return WM_ERR_ARG;
        pts += items[k].n_src + items[k].n_target;
    }
    if (pts == 0) return WM_OK;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    VoxelBatch vb;
    WM_TRY(vb.setup(ctx, items, idx, stride, mem));
    WM_TRY(vb.filter(leaf));
    size_t need = 0;
    for (unsigned c = 0; c < vb.n_clouds; ++c) {
        if (vb.n_out[c] == 0xFFFFFFFFu) return WM_ERR_ARG;  // leaf lattice beyond int32: wm_voxel_downsample returns the input
        n_out[c] = vb.n_out[c];
        need += vb.n_out[c];
    }
    if (need > cap_points) return WM_ERR_ARG;
    std::vector<float4> tmp;
    size_t w = 0;
    for (unsigned c = 0; c < vb.n_clouds; ++c) {
        if (!vb.n_out[c]) continue;
        tmp.resize(vb.n_out[c]);
        WM_HIP(ctx, hipMemcpy(tmp.data(), vb.filtered + vb.cl[c].off, vb.n_out[c] * sizeof(float4), hipMemcpyDeviceToHost));
        for (unsigned i = 0; i < vb.n_out[c]; ++i) {
            out_xyz[3 * w] = tmp[i].x, out_xyz[3 * w + 1] = tmp[i].y, out_xyz[3 * w + 2] = tmp[i].z;
            ++w;
        }
    }
    return WM_OK;
}

}  // extern "C"
