// wm_wave.hpp -- wavefront-wide reduction of the ICP accumulators, shared by the correspondence kernel
// (wm_nn.hip) and the one-workgroup registrations (wm_small.hip).
#ifndef WM_WAVE_HPP
#define WM_WAVE_HPP

#include "wm_internal.hpp"

namespace wm {

// Wave reduction of kAcc doubles by recursive halving: at the step for lane bit M a lane keeps one
// half of its values and sends the other half to lane ^ M, so the 18 values cost 9+5+3+2+1+1 = 21
// exchanges instead of 18 x 6.  The order of the additions is fixed by the lane numbers: the sums
// are bit-reproducible.  Afterwards component k sits in v[0] of the one lane acc_comp_of_lane()
// names (bit 0 clear; the other lanes hold padding zeros).
template <int C, int M>
__device__ __forceinline__ void acc_halve(double (&v)[kAcc], unsigned lane) {
    constexpr int H = (C + 1) / 2;
    const bool up = (lane & (unsigned) M) != 0u;
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const double lo = v[i];
        const double hi = (H + i < C) ? v[H + i] : 0.0;
        const double send = up ? lo : hi, keep = up ? hi : lo;
        v[i] = keep + __shfl_xor(send, M);
    }
    if constexpr (M > 1) acc_halve<H, M / 2>(v, lane);
}
__device__ __forceinline__ int acc_comp_of_lane(unsigned lane) {
    // follows acc_halve's (static) array sizes 18 -> 9 -> 5 -> 3 -> 2 -> 1 -> 1; `valid` = how many
    // leading entries of this lane's array are real components (the rest is zero padding)
    int c = kAcc, base = 0, valid = kAcc;
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

}  // namespace wm

#endif  // WM_WAVE_HPP
