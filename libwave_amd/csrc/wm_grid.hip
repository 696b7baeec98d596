// wm_grid.hip -- building the HBM-resident spatial index that replaces PCL's FLANN
// kd-tree (pcl::Registration::initCompute -> KdTreeFLANN::setInputCloud, triggered
// by icp.align at wave_matching/src/icp.cpp:95,116,126):
//   pack (AoS stride -> float4 + index, non-finite filter)  -> bbox reduction
//   -> cell histogram -> exclusive scan -> scatter (counting sort by cell)
// Target levels are sorted x-fastest so that x-adjacent cells are contiguous in
// memory; the source cloud is counting-sorted by a Morton cell code so that a
// wavefront's 64 consecutive queries stay spatially compact under any rigid T.
#include <cstring>  // before rocprim (its headers use memcpy unqualified)

#include <rocprim/rocprim.hpp>

#include "wm_internal.hpp"
#include "wm_sort.hpp"

#include <math.h>
#include <vector>
#include <string.h>

namespace wm {

// ------------------------------------------------------------------ pack
__global__ void __launch_bounds__(kBlock) k_pack(const unsigned char *in, size_t n, size_t stride,
                                                  float4 *out) {
    size_t i = (size_t) blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float *p = reinterpret_cast<const float *>(in + i * stride);
    float x = p[0], y = p[1], z = p[2];
    if (!(isfinite(x) && isfinite(y) && isfinite(z))) {
        x = y = z = __builtin_nanf("");
    }
    out[i] = make_float4(x, y, z, __uint_as_float((unsigned) i));
}

// A HOST cloud in PINNED memory (hipHostMalloc / hipHostRegister: what a caller that cares about the upload hands over)
// can cross PCIe on a DMA engine while this thread does something else: the copy into the slot-1 staging buffer is
// started here, asynchronously, on the side stream; pack_cloud(..., slot 1, staged = true) later waits for it -- on the
// HOST, so that the caller's memory is not read after the call that was given it has returned -- and packs from there.
// false: not pinned (or no side stream): the caller takes the blocking path.
template <bool PACK>
__global__ void k_bbox(float4 *pts, size_t n, float *out, const unsigned char *raw, size_t rstride);  // (below)

bool upload_begin_async(wm_ctx *ctx, const void *pts, size_t bytes) {
    if (!ctx->side_stream || bytes == 0) return false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, pts) != hipSuccess) {  // (plain pageable memory is unknown to the runtime: an error, cleared here)
        (void) hipGetLastError();
        return false;
    }
    if (at.type != hipMemoryTypeHost) return false;
    DevBuf &stage = ctx->staging2;
    if (stage.cap < bytes) {  // (growing it frees the old one: nothing of ours may still read it)
        if (hipStreamSynchronize(ctx->stream) != hipSuccess || stage.reserve(bytes) != hipSuccess) return false;
    }
    if (hipMemcpyAsync(stage.p, pts, bytes, hipMemcpyHostToDevice, ctx->side_stream) != hipSuccess) {
        (void) hipGetLastError();
        (void) hipStreamSynchronize(ctx->side_stream);
        return false;
    }
    return true;
}

int pack_cloud(wm_ctx *ctx, const void *pts, size_t n, size_t stride, int mem, float4 *out, int slot, bool staged,
               float *bbox_partials, unsigned *bbox_blocks) {
    if (bbox_blocks) {
        unsigned b = (unsigned) ((n + kBlock - 1) / kBlock);
        *bbox_blocks = b > (unsigned) kBboxBlocks ? (unsigned) kBboxBlocks : b;
    }
    if (n == 0) return WM_OK;
    const unsigned char *dptr = nullptr;
    if (mem == WM_MEM_HOST && staged) {
        // (upload_begin_async put the cloud on its way into the slot-1 staging buffer: wait for the copy engine)
        WM_HIP(ctx, hipStreamSynchronize(ctx->side_stream));
        dptr = ctx->staging2.as<unsigned char>();
    } else if (mem == WM_MEM_HOST) {
        // Caller memory is pageable: a blocking copy.  Slot 0 (the source, and every other caller): after the
        // stream has drained -- the staging buffer may still feed the previous cloud's k_pack.  Slot 1 (the
        // target, when the source of the same registration was uploaded just before): a staging buffer of
        // its own and NO wait -- the source's bounding box, Morton sort and gather run on the stream while
        // this cloud crosses PCIe (wm_set_target).
        DevBuf &stage = slot == 1 ? ctx->staging2 : ctx->staging;
        if (slot != 1) WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (stage.cap < n * stride) {  // (growing it frees the old one: nothing of ours may still read it)
            WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
            WM_HIP(ctx, stage.reserve(n * stride));
        }
        WM_HIP(ctx, hipMemcpy(stage.p, pts, n * stride, hipMemcpyHostToDevice));
        dptr = stage.as<unsigned char>();
    } else {
        dptr = static_cast<const unsigned char *>(pts);
    }
    unsigned blocks = (unsigned) ((n + kBlock - 1) / kBlock);
    if (bbox_partials && bbox_blocks && ctx->tune_pack_bbox) {  // (the cloud's box in the same launch: k_bbox<true>)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bbox<true>), dim3(*bbox_blocks), dim3(kBlock), 0, ctx->stream, out, n, bbox_partials, dptr, stride);
        WM_HIP(ctx, hipGetLastError());
        return WM_OK;
    }
    hipLaunchKernelGGL(k_pack, dim3(blocks), dim3(kBlock), 0, ctx->stream, dptr, n, stride, out);
    WM_HIP(ctx, hipGetLastError());
    if (bbox_partials && bbox_blocks) return launch_bbox(ctx, out, n, bbox_partials, bbox_blocks);
    return WM_OK;
}

// ------------------------------------------------------------------ bbox
// One partial per workgroup: out[8 b + 0..2] = min, [3..5] = max (as floats), [6] = valid count.
// The host reduces the partials (it waits for the result anyway); no same-address atomics.
// PACK: the cloud is packed on the way (k_pack's conversion of the caller's records at `raw`, byte stride `rstride`, into
// pts) -- one launch instead of two at the head of every registration's chain of dependent launches; the box and the count
// are the same numbers either way (minima, maxima and an integer do not care about the order they are formed in).
template <bool PACK>
__global__ void __launch_bounds__(kBlock) k_bbox(float4 *pts, size_t n, float *out, const unsigned char *raw, size_t rstride) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    unsigned cnt = 0;
    const size_t stride = (size_t) gridDim.x * kBlock;
    for (size_t i0 = (size_t) blockIdx.x * kBlock + threadIdx.x; i0 < n; i0 += 4 * stride) {
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // four loads in flight
            const size_t i = i0 + u * stride;
            if (PACK) {
                p[u] = make_float4(NAN, 0.f, 0.f, 0.f);
                if (i < n) {
                    const float *q = reinterpret_cast<const float *>(raw + i * rstride);
                    float x = q[0], y = q[1], z = q[2];
                    if (!(isfinite(x) && isfinite(y) && isfinite(z))) x = y = z = __builtin_nanf("");
                    p[u] = make_float4(x, y, z, __uint_as_float((unsigned) i));
                }
            } else {
                p[u] = i < n ? pts[i] : make_float4(NAN, 0.f, 0.f, 0.f);
            }
        }
        if (PACK) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t i = i0 + u * stride;
                if (i < n) pts[i] = p[u];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (p[u].x == p[u].x) {  // not NaN
                lo[0] = fminf(lo[0], p[u].x);
                lo[1] = fminf(lo[1], p[u].y);
                lo[2] = fminf(lo[2], p[u].z);
                hi[0] = fmaxf(hi[0], p[u].x);
                hi[1] = fmaxf(hi[1], p[u].y);
                hi[2] = fmaxf(hi[2], p[u].z);
                ++cnt;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        for (int d = 0; d < 3; ++d) {
            lo[d] = fminf(lo[d], __shfl_down(lo[d], off));
            hi[d] = fmaxf(hi[d], __shfl_down(hi[d], off));
        }
        cnt += __shfl_down(cnt, off);
    }
    __shared__ float s_lo[kBlock / 64][3], s_hi[kBlock / 64][3];
    __shared__ unsigned s_cnt[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        for (int d = 0; d < 3; ++d) {
            s_lo[wave][d] = lo[d];
            s_hi[wave][d] = hi[d];
        }
        s_cnt[wave] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) {
            for (int d = 0; d < 3; ++d) {
                lo[d] = fminf(lo[d], s_lo[w][d]);
                hi[d] = fmaxf(hi[d], s_hi[w][d]);
            }
            cnt += s_cnt[w];
        }
        float *o = out + 8 * (size_t) blockIdx.x;
        for (int d = 0; d < 3; ++d) {
            o[d] = lo[d];
            o[3 + d] = hi[d];
        }
        o[6] = __uint_as_float(cnt);
    }
}

int launch_bbox(wm_ctx *ctx, const float4 *pts, size_t n, float *partials_dev, unsigned *blocks_out) {
    unsigned blocks = (unsigned) ((n + kBlock - 1) / kBlock);
    if (blocks > (unsigned) kBboxBlocks) blocks = kBboxBlocks;
    *blocks_out = blocks;
    if (n == 0) return WM_OK;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bbox<false>), dim3(blocks), dim3(kBlock), 0, ctx->stream, const_cast<float4 *>(pts), n, partials_dev,
                       (const unsigned char *) nullptr, (size_t) 0);
    WM_HIP(ctx, hipGetLastError());
    return WM_OK;
}

void finish_bbox(const float *res, unsigned blocks, Bbox *out, size_t *n_valid) {
    for (int d = 0; d < 3; ++d) out->lo[d] = out->hi[d] = 0.f;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    size_t cnt = 0;
    for (unsigned b = 0; b < blocks; ++b) {
        const float *o = res + 8 * b;
        for (int d = 0; d < 3; ++d) {
            lo[d] = fminf(lo[d], o[d]);
            hi[d] = fmaxf(hi[d], o[3 + d]);
        }
        unsigned c;
        memcpy(&c, &o[6], 4);
        cnt += c;
    }
    *n_valid = cnt;
    if (cnt > 0)
        for (int d = 0; d < 3; ++d) {
            out->lo[d] = lo[d];
            out->hi[d] = hi[d];
        }
}

int compute_bbox(wm_ctx *ctx, const float4 *pts, size_t n, Bbox *out, size_t *n_valid) {
    for (int d = 0; d < 3; ++d) out->lo[d] = out->hi[d] = 0.f;
    *n_valid = 0;
    if (n == 0) return WM_OK;
    // partials in device memory, fetched into pinned memory by one wavefront (fast_fetch)
    float *res = (float *) pinned_scratch(ctx, 8 * sizeof(float) * kBboxBlocks);
    if (!res) return WM_ERR_HIP;
    WM_HIP(ctx, ctx->bbox_buf.reserve(8 * sizeof(float) * kBboxBlocks));
    unsigned blocks = 0;
    WM_TRY(launch_bbox(ctx, pts, n, ctx->bbox_buf.as<float>(), &blocks));
    WM_TRY(fast_fetch(ctx, res, ctx->bbox_buf.p, 8 * sizeof(float) * blocks));
    finish_bbox(res, blocks, out, n_valid);
    return WM_OK;
}

// -------------------------------------------------- counting sort by cell
struct LinearKey {  // x-fastest dense cell index (target levels)
    float ox, oy, oz, inv_h;
    int nx, ny, nz;
    __device__ unsigned operator()(const float4 &p) const {
        int cx = (int) floorf((p.x - ox) * inv_h);
        int cy = (int) floorf((p.y - oy) * inv_h);
        int cz = (int) floorf((p.z - oz) * inv_h);
        cx = min(max(cx, 0), nx - 1);
        cy = min(max(cy, 0), ny - 1);
        cz = min(max(cz, 0), nz - 1);
        return (unsigned) ((cz * ny + cy) * nx + cx);
    }
};

__device__ inline unsigned spread3(unsigned v) {  // 10 bits -> every third bit
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

struct MortonKey {  // Morton code of a 2^bits-per-axis cell (source ordering)
    float ox, oy, oz, sx, sy, sz;
    int cells;  // 2^bits
    __device__ unsigned operator()(const float4 &p) const {
        int cx = min(max((int) ((p.x - ox) * sx), 0), cells - 1);
        int cy = min(max((int) ((p.y - oy) * sy), 0), cells - 1);
        int cz = min(max((int) ((p.z - oz) * sz), 0), cells - 1);
        return spread3((unsigned) cx) | (spread3((unsigned) cy) << 1) |
               (spread3((unsigned) cz) << 2);
    }
};

// histogram of the cells; the value the atomic returns is the point's rank inside its cell, so the
// scatter needs no second round of atomics (the order inside a cell is the atomics' arrival order:
// arbitrary, and irrelevant -- the search's keys carry the original index)
template <class KeyFn>
__global__ void __launch_bounds__(kBlock) k_count(const float4 *pts, size_t n, KeyFn key,
                                                   unsigned *cell_of, unsigned *rank_of, unsigned *counts) {
    size_t i = (size_t) blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 p = pts[i];
    unsigned c = kNoIdx, r = 0;
    if (p.x == p.x) {
        c = key(p);
        r = atomicAdd(&counts[c], 1u);
    }
    cell_of[i] = c;
    rank_of[i] = r;
}

// (thread 0 also writes the four NaN entries behind the last point: the search reads groups of four
// entries and may run past the end of the last run -- see k_pad_tail, which does it for an empty level)
__global__ void __launch_bounds__(kBlock) k_scatter(const float4 *pts, size_t n,
                                                     const unsigned *cell_of, const unsigned *rank_of,
                                                     const unsigned *cell_start, float4 *out, size_t ncells) {
    size_t i = (size_t) blockIdx.x * kBlock + threadIdx.x;
    if (i == 0) {
        const unsigned end = cell_start[ncells];
        const float nanv = __builtin_nanf("");
#pragma unroll
        for (int u = 0; u < 4; ++u) out[end + u] = make_float4(nanv, nanv, nanv, __uint_as_float(kNoIdx));
    }
    if (i >= n) return;
    unsigned c = cell_of[i];
    if (c == kNoIdx) return;
    out[cell_start[c] + rank_of[i]] = pts[i];
}

// ------------------------------------------------------- exclusive scan
constexpr int kScanItems = 8;
constexpr int kScanTile = kBlock * kScanItems;  // 2048

__device__ inline unsigned block_exclusive_scan(unsigned v, unsigned *total, unsigned *lds) {
    // wave-level inclusive scan
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        unsigned t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    unsigned base = 0, tot = 0;
    for (int w = 0; w < kBlock / 64; ++w) {
        unsigned s = lds[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// pass 1: per-tile local exclusive scan (in place into out) + tile sums
__global__ void __launch_bounds__(kBlock) k_scan_tiles(const unsigned *in, size_t n, unsigned *out,
                                                        unsigned *tile_sums) {
    __shared__ unsigned lds[kBlock / 64];
    size_t base = (size_t) blockIdx.x * kScanTile + (size_t) threadIdx.x * kScanItems;
    unsigned v[kScanItems], sum = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        sum += v[k];
    }
    unsigned total;
    unsigned ex = block_exclusive_scan(sum, &total, lds);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k < n) out[base + k] = ex;
        ex += v[k];
    }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// pass 2: single block scans the tile sums (exclusive), carries across chunks
__global__ void __launch_bounds__(kBlock) k_scan_sums(unsigned *tile_sums, size_t ntiles,
                                                       unsigned *grand_total) {
    __shared__ unsigned lds[kBlock / 64];
    __shared__ unsigned carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (size_t c0 = 0; c0 < ntiles; c0 += kBlock) {
        size_t i = c0 + threadIdx.x;
        unsigned v = (i < ntiles) ? tile_sums[i] : 0u;
        unsigned total;
        unsigned ex = block_exclusive_scan(v, &total, lds);
        unsigned carry = carry_s;
        if (i < ntiles) tile_sums[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *grand_total = carry_s;
}

// pass 3: add tile offsets; also writes out[n] = grand total
__global__ void __launch_bounds__(kBlock) k_scan_add(unsigned *out, size_t n,
                                                      const unsigned *tile_sums,
                                                      const unsigned *grand_total) {
    size_t base = (size_t) blockIdx.x * kScanTile + (size_t) threadIdx.x * kScanItems;
    unsigned add = tile_sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < n) out[base + k] += add;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = *grand_total;
}

struct ScanIn {  // in[i] for i < n, 0 at i = n (exclusive_scan below)
    const unsigned *in;
    size_t n;
    __device__ unsigned operator()(size_t i) const { return i < n ? in[i] : 0u; }
};

// out must hold n+1 entries (out[n] = the total).  rocPRIM's single-pass look-back scan: 5 M cell counts in ~17 us against 41 us for the three-kernel
// tiles / tile sums / add scan above (kept for `WM_TUNE_SCAN=0`).
int exclusive_scan(wm_ctx *ctx, const unsigned *in, size_t n, unsigned *out) {
    if (n == 0) {
        WM_HIP(ctx, hipMemsetAsync(out, 0, sizeof(unsigned), ctx->stream));
        return WM_OK;
    }
    if (ctx->tune_scan) {
        // n + 1 items through an iterator that reads in[i] below n and 0 at n: out[n] comes out as the
        // total, without a one-thread kernel behind the scan (a dependent launch of its own: ~5 us)
        auto it = rocprim::make_transform_iterator(rocprim::counting_iterator<size_t>(0), ScanIn{in, n});
        // (32 items per thread: 54 us for the 14 M cell counts of a 1M-point level 0 against 67 with rocPRIM's default
        // tuning; 17 against 19 at 2 M -- scripts/dev/scan_probe.hip)
        using scan_cfg = rocprim::scan_config<256, 32, rocprim::block_load_method::block_load_transpose,
                                              rocprim::block_store_method::block_store_transpose,
                                              rocprim::block_scan_algorithm::using_warp_scan>;
        size_t bytes = 0;
        WM_HIP(ctx, rocprim::exclusive_scan<scan_cfg>(nullptr, bytes, it, out, 0u, n + 1, rocprim::plus<unsigned>(),
                                                      ctx->stream));
        WM_HIP(ctx, ctx->block_sums.reserve(bytes + 64));
        WM_HIP(ctx, rocprim::exclusive_scan<scan_cfg>(ctx->block_sums.p, bytes, it, out, 0u, n + 1,
                                                      rocprim::plus<unsigned>(), ctx->stream));
        WM_HIP(ctx, hipGetLastError());
        return WM_OK;
    }
    size_t ntiles = (n + kScanTile - 1) / kScanTile;
    WM_HIP(ctx, ctx->block_sums.reserve((ntiles + 2) * sizeof(unsigned)));
    unsigned *sums = ctx->block_sums.as<unsigned>();
    unsigned *grand = sums + ntiles;
    hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned) ntiles), dim3(kBlock), 0, ctx->stream, in, n,
                       out, sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kBlock), 0, ctx->stream, sums, ntiles, grand);
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned) ntiles), dim3(kBlock), 0, ctx->stream, out, n,
                       sums, grand);
    WM_HIP(ctx, hipGetLastError());
    return WM_OK;
}

template <class KeyFn>
static int counting_sort(wm_ctx *ctx, const float4 *pts, size_t n, KeyFn key, size_t ncells,
                         unsigned *cell_start /* ncells+1 */, float4 *out) {
    WM_HIP(ctx, ctx->cell_of.reserve(2 * n * sizeof(unsigned)));
    WM_HIP(ctx, ctx->counts.reserve(ncells * sizeof(unsigned)));
    unsigned *counts = ctx->counts.as<unsigned>();
    unsigned *cell_of = ctx->cell_of.as<unsigned>(), *rank_of = cell_of + n;
    WM_HIP(ctx, hipMemsetAsync(counts, 0, ncells * sizeof(unsigned), ctx->stream));
    unsigned blocks = (unsigned) ((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_count<KeyFn>), dim3(blocks), dim3(kBlock), 0, ctx->stream,
                       pts, n, key, cell_of, rank_of, counts);
    WM_HIP(ctx, hipGetLastError());
    WM_TRY(exclusive_scan(ctx, counts, ncells, cell_start));
    hipLaunchKernelGGL(k_scatter, dim3(blocks), dim3(kBlock), 0, ctx->stream, pts, n, cell_of, rank_of,
                       cell_start, out, ncells);
    WM_HIP(ctx, hipGetLastError());
    return WM_OK;
}

// occupied-cell count (for choosing the level-0 cell size): one partial per workgroup,
// summed by the host
constexpr int kOccBlocks = 1024;
__global__ void __launch_bounds__(kBlock) k_count_occupied(const unsigned *cell_start,
                                                            size_t ncells, unsigned *out) {
    unsigned c = 0;
    for (size_t i = (size_t) blockIdx.x * kBlock + threadIdx.x; i < ncells;
         i += (size_t) gridDim.x * kBlock)
        c += (cell_start[i + 1] != cell_start[i]);
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    __shared__ unsigned s_c[kBlock / 64];
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) c += s_c[w];
        out[blockIdx.x] = c;
    }
}

// Four NaN entries after the last point of a level's cell-sorted array: the search reads
// groups of four entries and may run past the end of the last run.
__global__ void k_pad_tail(const unsigned *cell_start, size_t ncells, float4 *pts) {
    const unsigned end = cell_start[ncells];
    const float nanv = __builtin_nanf("");
    if (threadIdx.x < 4) pts[end + threadIdx.x] = make_float4(nanv, nanv, nanv, __uint_as_float(kNoIdx));
}

static void grid_dims(const Bbox &bb, float h, int *nx, int *ny, int *nz) {
    *nx = (int) floor((bb.hi[0] - bb.lo[0]) / h) + 1;
    *ny = (int) floor((bb.hi[1] - bb.lo[1]) / h) + 1;
    *nz = (int) floor((bb.hi[2] - bb.lo[2]) / h) + 1;
}

int build_grid_level(wm_ctx *ctx, const float4 *pts, size_t n, const Bbox &bb, float h,
                     GridLevel *lvl, double *avg_occupancy) {
    int nx, ny, nz;
    grid_dims(bb, h, &nx, &ny, &nz);
    uint64_t ncells = (uint64_t) nx * ny * nz;
    lvl->ncells = ncells;
    WM_HIP(ctx, lvl->pts.reserve((n + 4) * sizeof(float4)));
    WM_HIP(ctx, lvl->cell_start.reserve((ncells + 1) * sizeof(unsigned)));
    LinearKey key{bb.lo[0], bb.lo[1], bb.lo[2], 1.0f / h, nx, ny, nz};
    if (n > 0) {
        WM_TRY(counting_sort(ctx, pts, n, key, ncells, lvl->cell_start.as<unsigned>(),
                             lvl->pts.as<float4>()));
    } else {
        WM_HIP(ctx, hipMemsetAsync(lvl->cell_start.p, 0, (ncells + 1) * sizeof(unsigned),
                                   ctx->stream));
        hipLaunchKernelGGL(k_pad_tail, dim3(1), dim3(64), 0, ctx->stream, lvl->cell_start.as<unsigned>(),
                           (size_t) ncells, lvl->pts.as<float4>());
    }
    // float cell assignment can be off by the rounding of (p - origin) * inv_h:
    // keep a cell-unit margin in every geometric bound that relies on it.
    float extent = fmaxf(fmaxf(bb.hi[0] - bb.lo[0], bb.hi[1] - bb.lo[1]), bb.hi[2] - bb.lo[2]);
    float amax = 0;
    for (int d = 0; d < 3; ++d) amax = fmaxf(amax, fmaxf(fabsf(bb.lo[d]), fabsf(bb.hi[d])));
    float ulp = fmaxf(amax, extent) * 1.2e-7f;
    lvl->d.ox = bb.lo[0];
    lvl->d.oy = bb.lo[1];
    lvl->d.oz = bb.lo[2];
    lvl->d.h = h;
    lvl->d.inv_h = 1.0f / h;
    lvl->d.slack = fmaxf(1e-3f, 8.0f * ulp / h);
    lvl->d.nx = nx;
    lvl->d.ny = ny;
    lvl->d.nz = nz;
    lvl->d.pts = lvl->pts.as<float4>();
    lvl->d.cell_start = lvl->cell_start.as<unsigned>();
    lvl->built = true;
    if (avg_occupancy) {
        unsigned blocks = (unsigned) ((ncells + kBlock - 1) / kBlock);
        if (blocks > (unsigned) kOccBlocks) blocks = kOccBlocks;
        unsigned *part = (unsigned *) pinned_scratch(ctx, sizeof(unsigned) * kOccBlocks);
        if (!part) return WM_ERR_HIP;
        WM_HIP(ctx, ctx->bbox_buf.reserve(sizeof(unsigned) * kOccBlocks));
        hipLaunchKernelGGL(k_count_occupied, dim3(blocks), dim3(kBlock), 0, ctx->stream,
                           lvl->cell_start.as<unsigned>(), (size_t) ncells, ctx->bbox_buf.as<unsigned>());
        WM_TRY(fast_fetch(ctx, part, ctx->bbox_buf.p, sizeof(unsigned) * blocks));
        uint64_t occ = 0;
        for (unsigned b = 0; b < blocks; ++b) occ += part[b];
        *avg_occupancy = occ ? (double) n / occ : 0.0;
    }
    return WM_OK;
}

// Source ordering: stable radix sort of (Morton cell code, original index) pairs, then a
// gather.  Stable => points of one Morton cell keep ascending original index, so every
// downstream sum over the source runs in a reproducible order.  Non-finite points get the
// key past the last cell and fall off the end (n_valid comes from the bbox pass).
__global__ void __launch_bounds__(kBlock) k_morton_keys(const float4 *pts, unsigned n, MortonKey key,
                                                         unsigned invalid, unsigned *keys,
                                                         unsigned *vals) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    keys[i] = p.x == p.x ? key(p) : invalid;
    vals[i] = i;
}

__global__ void __launch_bounds__(kBlock) k_gather(const float4 *pts, const unsigned *vals,
                                                    unsigned n, float4 *out) {
    const unsigned j = blockIdx.x * kBlock + threadIdx.x;
    if (j < n) out[j] = pts[vals[j]];
}

int morton_sort(wm_ctx *ctx, const float4 *pts, size_t n, const Bbox &bb, size_t n_valid,
                float4 *out) {
    if (n == 0 || n_valid == 0) return WM_OK;
    int bits = (int) ceil(log2(cbrt((double) n)));
    if (bits < 3) bits = 3;
    if (bits > 8) bits = 8;
    const int cells = 1 << bits;
    float ext[3];
    for (int d = 0; d < 3; ++d) ext[d] = fmaxf(bb.hi[d] - bb.lo[d], 1e-6f);
    MortonKey key{bb.lo[0], bb.lo[1], bb.lo[2], cells / ext[0], cells / ext[1], cells / ext[2],
                  cells};
    WM_HIP(ctx, ctx->vg_idx.reserve(n * 4));
    WM_HIP(ctx, ctx->vg_idx2.reserve(n * 4));
    WM_HIP(ctx, ctx->vg_perm.reserve(n * 4));
    WM_HIP(ctx, ctx->vg_perm2.reserve(n * 4));
    unsigned *k1 = ctx->vg_idx.as<unsigned>(), *k2 = ctx->vg_idx2.as<unsigned>();
    unsigned *v1 = ctx->vg_perm.as<unsigned>(), *v2 = ctx->vg_perm2.as<unsigned>();
    const unsigned blocks = (unsigned) ((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_morton_keys, dim3(blocks), dim3(kBlock), 0, ctx->stream, pts, (unsigned) n,
                       key, 1u << (3 * bits), k1, v1);
    size_t tmp_bytes = 0;
    WM_HIP(ctx, sort_pairs_low_bits(nullptr, tmp_bytes, k1, k2, v1, v2, n, 3 * bits + 1, ctx->stream,
                                    (size_t) ctx->tune_radix_min, ctx->tune_sort));
    WM_HIP(ctx, ctx->vg_tmp.reserve(tmp_bytes));
    WM_HIP(ctx, sort_pairs_low_bits(ctx->vg_tmp.p, tmp_bytes, k1, k2, v1, v2, n, 3 * bits + 1, ctx->stream,
                                    (size_t) ctx->tune_radix_min, ctx->tune_sort));
    const unsigned gblocks = (unsigned) ((n_valid + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_gather, dim3(gblocks), dim3(kBlock), 0, ctx->stream, pts, v2,
                       (unsigned) n_valid, out);
    WM_HIP(ctx, hipGetLastError());
    return WM_OK;
}

// ---- coarser levels derived from the next finer one (cell size exactly x2, same origin):
// a coarse cell is the union of 2x2x2 fine cells, i.e. of 4 contiguous runs of the fine
// cell-sorted array (the two x-children of a row are adjacent).  Counting is a handful of
// cell_start reads per coarse cell and the copy moves whole runs: no atomics, no keys.
__device__ __forceinline__ void child_run(const unsigned *fs, int fnx, int fny, int fnz, int X,
                                          int Y, int Z, int r, unsigned *s, unsigned *e) {
    const int y = 2 * Y + (r & 1), z = 2 * Z + (r >> 1);
    *s = *e = 0;
    if (y >= fny || z >= fnz) return;
    const size_t base = ((size_t) z * fny + y) * fnx;
    const int xa = 2 * X, xb = min(2 * X + 2, fnx);
    *s = fs[base + xa];
    *e = fs[base + xb];
}

__global__ void __launch_bounds__(kBlock)
    k_coarse_count(const unsigned *__restrict__ fs, int fnx, int fny, int fnz, int cnx, int cny,
                   size_t ncells, unsigned *__restrict__ counts) {
    const size_t c = (size_t) blockIdx.x * kBlock + threadIdx.x;
    if (c >= ncells) return;
    const int X = (int) (c % cnx), Y = (int) ((c / cnx) % cny), Z = (int) (c / ((size_t) cnx * cny));
    unsigned total = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        unsigned s, e;
        child_run(fs, fnx, fny, fnz, X, Y, Z, r, &s, &e);
        total += e - s;
    }
    counts[c] = total;
}

// One lane per POINT of the fine array: its level-0 cell (recomputed from its coordinates,
// exactly as the level-0 sort did) shifted right gives its fine and coarse cells; its slot in
// the coarse array is the coarse cell's start + the lengths of the child runs before its own
// + its offset in its run.  Reads and writes are coalesced (a run moves as a block) and the
// order inside a coarse cell is fixed: child runs in (y, z) order, fine order inside a run.
__global__ void __launch_bounds__(kBlock)
    k_coarse_move(const unsigned *__restrict__ fs, const float4 *__restrict__ fpts, size_t fncells,
                  LinearKey key0, int shift, int fnx, int fny, int fnz, int cnx, int cny,
                  const unsigned *__restrict__ cs, float4 *__restrict__ out, size_t cncells) {
    const unsigned j = blockIdx.x * kBlock + threadIdx.x;
    if (j == 0) {  // the four NaN entries behind the last point (see k_scatter)
        const unsigned end = cs[(size_t) cncells];
        const float nanv = __builtin_nanf("");
#pragma unroll
        for (int u = 0; u < 4; ++u) out[end + u] = make_float4(nanv, nanv, nanv, __uint_as_float(kNoIdx));
    }
    if (j >= fs[fncells]) return;
    const float4 p = fpts[j];
    int x = min(max((int) floorf((p.x - key0.ox) * key0.inv_h), 0), key0.nx - 1) >> shift;
    int y = min(max((int) floorf((p.y - key0.oy) * key0.inv_h), 0), key0.ny - 1) >> shift;
    int z = min(max((int) floorf((p.z - key0.oz) * key0.inv_h), 0), key0.nz - 1) >> shift;
    const int X = x >> 1, Y = y >> 1, Z = z >> 1, r = (y & 1) + 2 * (z & 1);
    unsigned dst = cs[((size_t) Z * cny + Y) * cnx + X];
    for (int q = 0; q < r; ++q) {
        unsigned s, e;
        child_run(fs, fnx, fny, fnz, X, Y, Z, q, &s, &e);
        dst += e - s;
    }
    unsigned s, e;
    child_run(fs, fnx, fny, fnz, X, Y, Z, r, &s, &e);
    out[dst + (j - s)] = p;
}

static int derive_grid_level(wm_ctx *ctx, const GridLevel &fine, int fine_index, GridLevel *lvl,
                             size_t n) {
    const GridDev &f = fine.d;
    const int nx = (f.nx + 1) / 2, ny = (f.ny + 1) / 2, nz = (f.nz + 1) / 2;
    const uint64_t ncells = (uint64_t) nx * ny * nz;
    lvl->ncells = ncells;
    WM_HIP(ctx, lvl->pts.reserve((n + 4) * sizeof(float4)));
    WM_HIP(ctx, lvl->cell_start.reserve((ncells + 1) * sizeof(unsigned)));
    WM_HIP(ctx, ctx->counts.reserve(ncells * sizeof(unsigned)));
    unsigned *counts = ctx->counts.as<unsigned>();
    const unsigned blocks = (unsigned) ((ncells + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_coarse_count, dim3(blocks), dim3(kBlock), 0, ctx->stream, f.cell_start, f.nx,
                       f.ny, f.nz, nx, ny, (size_t) ncells, counts);
    WM_TRY(exclusive_scan(ctx, counts, ncells, lvl->cell_start.as<unsigned>()));
    {
        const GridDev &g0 = ctx->levels[0].d;
        const LinearKey key0{g0.ox, g0.oy, g0.oz, g0.inv_h, g0.nx, g0.ny, g0.nz};
        const unsigned pblocks = (unsigned) ((n + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(k_coarse_move, dim3(pblocks), dim3(kBlock), 0, ctx->stream, f.cell_start,
                           f.pts, (size_t) fine.ncells, key0, fine_index, f.nx, f.ny, f.nz, nx, ny,
                           lvl->cell_start.as<unsigned>(), lvl->pts.as<float4>(), (size_t) ncells);
    }
    WM_HIP(ctx, hipGetLastError());
    lvl->d = f;
    lvl->d.h = 2.0f * f.h;  // exact in float
    lvl->d.inv_h = 1.0f / lvl->d.h;
    lvl->d.nx = nx;
    lvl->d.ny = ny;
    lvl->d.nz = nz;
    lvl->d.pts = lvl->pts.as<float4>();
    lvl->d.cell_start = lvl->cell_start.as<unsigned>();
    // a point sits in coarse cell (fine cell >> 1): the fine level's geometric margin
    // (slack_f * h_f) is slack_f / 2 coarse cells; keep the fine value (conservative)
    lvl->built = true;
    return WM_OK;
}

// Level 0: cell size from the measured density (a few points per occupied cell).
// Levels >= 1: cell size x4 per level until one ring of cells covers max_corr.
static int build_level0(wm_ctx *ctx) {
    const float4 *pts = ctx->tgt_orig.as<float4>();
    const Bbox &bb = ctx->tgt_bbox;
    const size_t n = ctx->n_tgt_input;
    double ext[3], vol = 1;
    for (int d = 0; d < 3; ++d) {
        ext[d] = fmax((double) bb.hi[d] - bb.lo[d], 1e-3);
        vol *= ext[d];
    }
    const uint64_t kMaxCells = 96ull << 20;
    auto clamp_h = [&](double h) {
        for (;;) {
            int nx, ny, nz;
            grid_dims(bb, (float) h, &nx, &ny, &nz);
            if ((uint64_t) nx * ny * nz <= kMaxCells) return h;
            h *= 1.26;
        }
    };
    double h;
    if (ctx->grid_cell_override > 0) {
        h = clamp_h(ctx->grid_cell_override);
        return build_grid_level(ctx, pts, n, bb, (float) h, &ctx->levels[0], nullptr);
    }
    const double target_occ = 3.0;
    // first guess assumes points spread over surfaces: occupancy ~ h^2 ...
    h = clamp_h(fmax(cbrt(vol / fmax((double) ctx->n_tgt, 1.0)) * 1.5, 1e-4));
    // ... unless the previous target looked alike (consecutive scans of one sensor do):
    // then its tuned cell size is the first guess and usually passes the check at once
    if (ctx->tuned_h > 0 && ctx->tuned_n > 0) {
        const double rn = (double) ctx->n_tgt / (double) ctx->tuned_n;
        const double rv = vol / ctx->tuned_vol;
        if (rn > 0.8 && rn < 1.25 && rv > 0.6 && rv < 1.6) h = clamp_h(ctx->tuned_h);
        // a cloud this close to the one the cell size was measured on: build with it and skip the
        // occupancy check (a device -> host round trip) -- but look again every 16th time.  The cell
        // size only steers the search's work, never its result.
        if (rn > 0.9 && rn < 1.1 && rv > 0.8 && rv < 1.25 && ++ctx->tuned_uses < 16)
            return build_grid_level(ctx, pts, n, bb, (float) h, &ctx->levels[0], nullptr);
    }
    ctx->tuned_uses = 0;
    double occ = 0;
    for (int it = 0; it < 3; ++it) {
        WM_TRY(build_grid_level(ctx, pts, n, bb, (float) h, &ctx->levels[0], &occ));
        if (occ <= 0) break;
        if (occ > target_occ * 0.6 && occ < target_occ * 1.6) break;
        double scale = sqrt(target_occ / occ);
        if (scale > 4) scale = 4;
        if (scale < 0.25) scale = 0.25;
        double nh = clamp_h(h * scale);
        if (fabs(nh - h) < 0.05 * h) break;
        h = nh;
    }
    if (!ctx->levels[0].built || ctx->levels[0].d.h != (float) h)
        WM_TRY(build_grid_level(ctx, pts, n, bb, (float) h, &ctx->levels[0], nullptr));
    ctx->tuned_h = h;
    ctx->tuned_n = ctx->n_tgt;
    ctx->tuned_vol = vol;
    return WM_OK;
}

int ensure_levels(wm_ctx *ctx, double max_corr) {
    if (!ctx->levels[0].built) {
        WM_TRY(build_level0(ctx));
        ctx->n_levels = 1;
        ctx->levels_max_corr = -1;
    }
    if (ctx->levels_max_corr == max_corr) return WM_OK;
    const float4 *pts = ctx->tgt_orig.as<float4>();
    double h = ctx->levels[0].d.h;
    int L = 1;
    // cell size x2 per level until a level's cells reach max_corr / 5: the search picks the
    // finest level whose cells are >= 0.2 r (lane scan) or 0.5 r (cooperative scan) and walks
    // boxes of cells, so coarser levels than that would never be chosen by the former and save
    // the latter (rare far-out queries) only a few row look-ups
    while (h < 0.2 * max_corr && L < kMaxLevels) {
        h *= 2.0;
        WM_TRY(derive_grid_level(ctx, ctx->levels[L - 1], L - 1, &ctx->levels[L], ctx->n_tgt_input));
        ++L;
    }
    (void) pts;
    ctx->n_levels = L;
    ctx->levels_max_corr = max_corr;
    // publish the ladder for the search kernel
    LevelsDev host{};
    for (int l = 0; l < L; ++l) host.g[l] = ctx->levels[l].d;
    host.n = L;
    WM_HIP(ctx, ctx->d_levels.reserve(sizeof(LevelsDev)));
    // staged in pinned memory the ctx owns: the copy is asynchronous and needs no wait (the
    // next user of the scratch waits on the stream before touching it)
    LevelsDev *stage = (LevelsDev *) ((char *) pinned_scratch(ctx, 0) + (32u << 10));
    if (!ctx->h_scratch) return WM_ERR_HIP;
    *stage = host;
    WM_HIP(ctx, hipMemcpyAsync(ctx->d_levels.p, stage, sizeof(host), hipMemcpyHostToDevice,
                               ctx->stream));
    return WM_OK;
}

}  // namespace wm

extern "C" int wm_debug_sort_pairs(wm_ctx *ctx, const void *keys, int key_bytes, size_t n, unsigned bits, unsigned *values_out) {
    using namespace wm;
    if (!ctx || (n > 0 && (!keys || !values_out)) || (key_bytes != 4 && key_bytes != 8) || bits > 8u * (unsigned) key_bytes ||
        n >= (size_t) 0xFFFF0000u)
        return WM_ERR_ARG;
    if (n == 0) return WM_OK;
    WM_HIP(ctx, hipSetDevice(ctx->device));
    struct Scoped : DevBuf {  // (DevBuf frees nothing by itself: the context owns its buffers; these are this call's)
        ~Scoped() { release(); }
    } k1, k2, v1, v2, tmp;
    WM_HIP(ctx, k1.reserve(n * (size_t) key_bytes));
    WM_HIP(ctx, k2.reserve(n * (size_t) key_bytes));
    WM_HIP(ctx, v1.reserve(n * 4));
    WM_HIP(ctx, v2.reserve(n * 4));
    WM_HIP(ctx, tmp.reserve(rs_temp_bytes(n)));
    std::vector<unsigned> iota(n);
    for (size_t i = 0; i < n; ++i) iota[i] = (unsigned) i;
    WM_HIP(ctx, hipMemcpy(k1.p, keys, n * (size_t) key_bytes, hipMemcpyHostToDevice));
    WM_HIP(ctx, hipMemcpy(v1.p, iota.data(), n * 4, hipMemcpyHostToDevice));
    if (key_bytes == 4)
        WM_HIP(ctx, rs_sort_pairs(tmp.p, k1.as<unsigned>(), k2.as<unsigned>(), v1.as<unsigned>(), v2.as<unsigned>(), n, bits, ctx->stream));
    else
        WM_HIP(ctx, rs_sort_pairs(tmp.p, k1.as<unsigned long long>(), k2.as<unsigned long long>(), v1.as<unsigned>(), v2.as<unsigned>(), n,
                                  bits, ctx->stream));
    WM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    WM_HIP(ctx, hipMemcpy(values_out, v2.p, n * 4, hipMemcpyDeviceToHost));
    return WM_OK;
}
