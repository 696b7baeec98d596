// wm_wave.hpp -- wavefront-wide reduction of the ICP accumulators, shared by the correspondence kernel
// (wm_nn.hip) and the one-workgroup registrations (wm_small.hip).
#ifndef WM_WAVE_HPP
#define WM_WAVE_HPP

#include "wm_internal.hpp"

// v_permlane32_swap / v_permlane16_swap exist on gfx950 only, and this library is written for gfx950
// only (CMakeLists.txt and the Makefile pin it): say so instead of failing inside a builtin.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "wavematch-hip targets gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif

namespace wm {

// Wave reduction of kAcc doubles by recursive halving: at the step for lane bit M a lane keeps one
// half of its values and sends the other half to lane ^ M, so the 18 values cost 9+5+3+2+1+1 = 21
// exchanges instead of 18 x 6.  The order of the additions is fixed by the lane numbers: the sums
// are bit-reproducible.  Afterwards component k sits in v[0] of the one lane acc_comp_of_lane()
// names (bit 0 clear; the other lanes hold padding zeros).
// The two coarse steps (lane bits 32 and 16: 14 of the 21 exchanges) use gfx950's lane-swap
// instructions: v_permlane32_swap exchanges lanes 32-63 of one register with lanes 0-31 of another,
// v_permlane16_swap the odd 16-lane rows of one with the even rows of the other -- after swapping the
// "low" component with the "high" one, their sum IS keep + received in every lane: no select, no trip
// through the LDS crossbar, the same additions (bit-identical sums).
template <int C, int M>
__device__ __forceinline__ void acc_halve(double (&v)[kAcc], unsigned lane) {
    constexpr int H = (C + 1) / 2;
    if constexpr (M == 32 || M == 16) {
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const double lo = v[i];
            const double hi = (H + i < C) ? v[H + i] : 0.0;
            const unsigned l0 = (unsigned) __double2loint(lo), l1 = (unsigned) __double2hiint(lo);
            const unsigned h0 = (unsigned) __double2loint(hi), h1 = (unsigned) __double2hiint(hi);
            unsigned a0, a1, b0, b1;
            if constexpr (M == 32) {
                const auto r0 = __builtin_amdgcn_permlane32_swap(l0, h0, false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(l1, h1, false, false);
                a0 = r0[0], b0 = r0[1], a1 = r1[0], b1 = r1[1];
            } else {
                const auto r0 = __builtin_amdgcn_permlane16_swap(l0, h0, false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(l1, h1, false, false);
                a0 = r0[0], b0 = r0[1], a1 = r1[0], b1 = r1[1];
            }
            v[i] = __hiloint2double((int) a1, (int) a0) + __hiloint2double((int) b1, (int) b0);
        }
    } else {
        const bool up = (lane & (unsigned) M) != 0u;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const double lo = v[i];
            const double hi = (H + i < C) ? v[H + i] : 0.0;
            const double send = up ? lo : hi, keep = up ? hi : lo;
            v[i] = keep + __shfl_xor(send, M);
        }
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
