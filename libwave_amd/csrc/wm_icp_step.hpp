// wm_icp_step.hpp -- what one ICP iteration does with its statistics: the compact accumulators of
// the correspondence pass expanded to the public layout, then the solve and PCL's stopping rules
// (pcl::IterativeClosestPoint::computeTransformation + DefaultConvergenceCriteria, called from
// wave_matching/src/icp.cpp:95,116,126).  Shared by the per-iteration solve kernel (wm_icp.hip), the
// sharded path's host step and the one-workgroup registrations of wm_small.hip.
#ifndef WM_ICP_STEP_HPP
#define WM_ICP_STEP_HPP

#include "wm_internal.hpp"

namespace wm {

// expand the 17 compact accumulators to the public 32-slot layout
// every (mask + 1)-th query reports whether its match changed: mask = 2^s - 1 with the smallest s for which
// n >> s < 2^23 (0 for every cloud below 8.4M points)
__device__ __host__ inline unsigned changed_mask_for(unsigned n) {
    unsigned s = 0;
    while ((n >> s) >= (1u << 23)) ++s;
    return (1u << s) - 1u;
}

__device__ __host__ inline void expand_stats(int mode, const double *a, double *st, unsigned changed_mask = 0u) {
#pragma unroll
    for (int k = 0; k < kStatsLen; ++k) st[k] = 0.0;
    // source points handled (ownership check of the sharded path), and -- the fraction of a[17], in
    // units of 2^-24 -- how many of them changed their match (a free slot of either layout)
    st[kStatsLen - 1] = floor(a[17]);
    st[kStatsLen - 3] = (a[17] - floor(a[17])) * 16777216.0 * (double) (changed_mask + 1u);
    if (mode == WM_ICP_SVD) {
        st[kSvdN] = a[0];
        for (int k = 0; k < 3; ++k) st[kSvdSp + k] = a[1 + k];
        for (int k = 0; k < 3; ++k) st[kSvdSq + k] = a[4 + k];
        for (int k = 0; k < 9; ++k) st[kSvdSqp + k] = a[7 + k];
        st[kSvdSd2] = a[16];
    } else {
        const double n = a[0], sx = a[1], sy = a[2], sz = a[3];
        double H[36];
#pragma unroll
        for (int k = 0; k < 36; ++k) H[k] = 0;
        H[0] = H[7] = H[14] = n;
        H[0 * 6 + 4] = sz;
        H[0 * 6 + 5] = -sy;
        H[1 * 6 + 3] = -sz;
        H[1 * 6 + 5] = sx;
        H[2 * 6 + 3] = sy;
        H[2 * 6 + 4] = -sx;
        H[3 * 6 + 3] = a[4];
        H[3 * 6 + 4] = a[5];
        H[3 * 6 + 5] = a[6];
        H[4 * 6 + 4] = a[7];
        H[4 * 6 + 5] = a[8];
        H[5 * 6 + 5] = a[9];
        st[kGnN] = n;
        st[kGnSd2] = a[16];
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (j >= i) st[kGnH + k++] = H[i * 6 + j];
#pragma unroll
        for (int i = 0; i < 6; ++i) st[kGnG + i] = a[10 + i];
    }
}

// The solve + stopping rules of one ICP iteration, from the (all-reduced) statistics.
// Runs as lane 0 of k_reduce_solve on the GPU and as plain host code in
// wm_host_icp_apply (the sharded path's CPU tests drive exactly this function).
// uns_total >= 0: the sum of st->cert_unsettled[] (already added up, and zeroed, by the caller's other lanes)
__host__ __device__ inline void icp_apply_stats(IcpDevState *st, const double *stats_in, long long uns_total = -1) {

    // work on a register copy: every st-> access is a global round trip
    double stats[kStatsLen];
#pragma unroll
    for (int k = 0; k < kStatsLen; ++k) stats[k] = stats_in[k];
    const int mode = st->mode;
#pragma unroll
    for (int k = 0; k < 12; ++k) st->Tf_search[k] = st->Tf[k];  // (the pose these statistics were searched under)
    const double n = stats[0];
    const double sd2 = mode == WM_ICP_SVD ? stats[kSvdSd2] : stats[kGnSd2];
    const double mse = n > 0 ? sd2 / n : 0.0;
    st->n_corr = (int) n;
    st->mse = mse;
    {
        const double expect = st->expect_owned < 0 ? stats[kStatsLen - 2] : st->expect_owned;
        if (st->expect_owned != 0 && stats[kStatsLen - 1] != expect) st->owned_violations += 1;
    }
    // what the host steers the choice of the next search kernel by
    {
        const double handled = stats[kStatsLen - 1];
        unsigned uns = 0;
        if (uns_total >= 0) {
            uns = (unsigned) uns_total;
        } else {
#pragma unroll 1
            for (int k = 0; k < 64; ++k) {
                uns += st->cert_unsettled[k];
                st->cert_unsettled[k] = 0u;
            }
        }
        // (sharded: `handled` is the all-reduced count, the searches counted are this rank's own)
        const double mine = (st->local_handled > 0 && !st->uns_global) ? st->local_handled : handled;
        st->frac_changed = handled > 0 ? (float) (stats[kStatsLen - 3] / handled) : 0.f;
        st->frac_unsettled = mine > 0 ? (float) ((double) uns / mine) : 0.f;  // (0 after a full search: nothing counted)
    }
    // bookkeeping for the next iteration's queues
    st->deferred_total += st->queue_count[1];
#pragma unroll
    for (int l = 0; l <= kMaxLevels; ++l) st->queue_count[l] = 0;
    if (n < 3.0) {  // PCL: min_number_correspondences_ = 3
        st->state = WM_CONV_NO_CORRESPONDENCES;
        st->converged = 0;
        st->done = 1;
        return;
    }
    double Tk[16], Tc[16], Tn[16];
#ifdef __HIP_DEVICE_COMPILE__
    st->dbg[4] = clock64();
#endif
    if (mode == WM_ICP_SVD)
        umeyama_from_stats(stats, Tk, st->svd_warm ? st->svd_v : nullptr);
    else
        gn6_from_stats(stats, Tk);
#ifdef __HIP_DEVICE_COMPILE__
    st->dbg[5] = clock64();
#endif
#pragma unroll
    for (int k = 0; k < 16; ++k) Tc[k] = st->T[k];
    mat4_mul(Tk, Tc, Tn);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        st->T[k] = Tn[k];
        st->Tk[k] = Tk[k];
    }
    // how far this step moves the source points at most: |(R_k - I) c + t_k| + |R_k - I|_F rho, with c
    // and rho the centre and half diagonal of the cloud under the pose the step was computed for
    // (what decides whether the next searches can be certified instead: k_nn_cert)
    {
        const double c0x = st->src_centre[0], c0y = st->src_centre[1], c0z = st->src_centre[2];
        const double cx = Tc[0] * c0x + Tc[1] * c0y + Tc[2] * c0z + Tc[3];
        const double cy = Tc[4] * c0x + Tc[5] * c0y + Tc[6] * c0z + Tc[7];
        const double cz = Tc[8] * c0x + Tc[9] * c0y + Tc[10] * c0z + Tc[11];
        const double a0 = Tk[0] - 1.0, a5 = Tk[5] - 1.0, a10 = Tk[10] - 1.0;
        const double dx = a0 * cx + Tk[1] * cy + Tk[2] * cz + Tk[3];
        const double dy = Tk[4] * cx + a5 * cy + Tk[6] * cz + Tk[7];
        const double dz = Tk[8] * cx + Tk[9] * cy + a10 * cz + Tk[11];
        const double fro2 = a0 * a0 + Tk[1] * Tk[1] + Tk[2] * Tk[2] + Tk[4] * Tk[4] + a5 * a5 + Tk[6] * Tk[6] +
                            Tk[8] * Tk[8] + Tk[9] * Tk[9] + a10 * a10;
        // (a heuristic's input: float square roots, not two one-lane f64 chains)
        st->step_disp = sqrtf((float) (dx * dx + dy * dy + dz * dz)) + sqrtf((float) fro2) * st->src_radius;
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) st->Tf[k] = (float) Tn[k];
    const int iter = st->iter + 1;
    st->iter = iter;
    st->have_prev = 1;
    const int max_iter = st->max_iter;

    // pcl::registration::DefaultConvergenceCriteria::hasConverged()
    if (st->forced) {
        if (iter >= max_iter) {
            st->converged = 1;
            st->state = WM_CONV_FORCED;
            st->done = 1;
        }
        st->prev_mse = mse;
        return;
    }
    if (iter >= max_iter) {
        st->converged = 1;
        st->state = WM_CONV_ITERATIONS;
        st->done = 1;
        return;
    }
    const double prev_mse = st->prev_mse;
    const double cos_angle = 0.5 * (Tk[0] + Tk[5] + Tk[10] - 1.0);
    const double tsq = Tk[3] * Tk[3] + Tk[7] * Tk[7] + Tk[11] * Tk[11];
    int state = WM_CONV_NOT_CONVERGED;
    if (cos_angle >= st->rot_thr && tsq <= st->trans_thr)
        state = WM_CONV_TRANSFORM;
    else if (fabs(mse - prev_mse) < 1e-12)
        state = WM_CONV_ABS_MSE;
    else if (fabs(mse - prev_mse) / prev_mse < st->fit_eps)
        state = WM_CONV_REL_MSE;
    if (state != WM_CONV_NOT_CONVERGED) {
        st->converged = 1;
        st->state = state;
        st->done = 1;
        return;
    }
    st->prev_mse = mse;
}

}  // namespace wm

#endif  // WM_ICP_STEP_HPP
