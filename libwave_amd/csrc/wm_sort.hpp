// wm_sort.hpp -- the library sort used by the voxel grids and the Morton order: rocPRIM's stable
// LSD radix sort of (key, value) pairs over the low `bits` bits of the key.
//
// rocPRIM's default dispatch takes its MERGE sort for up to 1 M items (log2(n / 4096) merge
// passes of two kernels each, whatever the key width): eight passes for a 1 M-point cloud.  The
// keys here are narrow (17-30 bits: leaf / Morton indices), so the onesweep radix path needs
// two to four passes; above `radix_min` items (256 k by default: 7 % of a default multiscale match at
// 1 M; at 500 k the merge path's 21 launches take the host longer to enqueue -- ~190 us -- than the
// GPU needs for either; slightly worse at 100 k) it is selected explicitly (merge-sort limit 0).  Both paths are stable, so the result does not depend on the choice.
#ifndef WM_SORT_HPP
#define WM_SORT_HPP

#include <rocprim/rocprim.hpp>

namespace wm {

template <class K, class V>
inline hipError_t sort_pairs_low_bits(void *tmp, size_t &tmp_bytes, K *keys_in, K *keys_out, V *vals_in,
                                      V *vals_out, size_t n, unsigned bits, hipStream_t stream,
                                      size_t radix_min = (size_t) (256u << 10)) {
    using radix_only = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                  rocprim::default_config, 0>;
    if (n > radix_min)
        return rocprim::radix_sort_pairs<radix_only>(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n,
                                                     0u, bits, stream);
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, bits, stream);
}

}  // namespace wm

#endif  // WM_SORT_HPP
