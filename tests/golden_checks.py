"""Shared assertions of an implementation (the C oracle on the CPU, the HIP path on the GPU) against
tests/golden/gicp_ndt_golden.json -- the independent numpy / scipy restatement of PCL's NDT and GICP
arithmetic on the reference's fixture (tests/golden/make_golden_gicp_ndt.py)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def golden():
    with open(os.path.join(HERE, "golden", "gicp_ndt_golden.json")) as f:
        return json.load(f)


def shifted(scan, tx):
    P = np.eye(4)
    P[0, 3] = tx
    return (scan.astype(np.float64) @ P[:3, :3].T + P[:3, 3]).astype(np.float32), P


def pose_err(A, B):
    dt = float(np.linalg.norm(np.asarray(A)[:3, 3] - np.asarray(B)[:3, 3]))
    R = np.asarray(A)[:3, :3].T @ np.asarray(B)[:3, :3]
    return dt, float(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1)))


def check_ndt_derivatives(case, derivs):
    """derivs(pose, pcl_d1_sign) -> (score, grad[6], hess[6,6], n_voxels).  The golden values use a
    float transform built from double trigonometry; PCL (and both implementations) build it from
    float trigonometry: the moved points differ by an ulp of float, hence the 2e-5 bars."""
    for ev in case["evals"]:
        pose = np.array(ev["pose"])
        s, g, H, nv = derivs(pose, 0)
        assert nv == case["n_voxels"]
        assert abs(s - ev["score"]) <= 2e-5 * abs(ev["score"]), (s, ev["score"])
        G, HH = np.array(ev["grad"]), np.array(ev["hess"])
        np.testing.assert_allclose(g, G, rtol=2e-4, atol=2e-5 * np.abs(G).max())
        np.testing.assert_allclose(H, HH, rtol=2e-4, atol=2e-5 * np.abs(HH).max())
        # PCL's h_ang d1[2] = +sy (true: -sy) touches H(4,4) only, by this much
        s1, g1, H1, _ = derivs(pose, 1)
        D = H1 - H
        want = ev["hess44_pcl_minus_true"]
        assert abs(D[4, 4] - want) <= 2e-4 * max(abs(want), 1e-9 * np.abs(HH).max()) + 1e-9 * np.abs(HH).max()
        D[4, 4] = 0
        assert np.abs(D).max() <= 1e-9 * np.abs(HH).max() and s1 == s and np.array_equal(g1, g)


def noisy_filtered_pair(scan, case):
    """The pair of GOLD["gicp"]["noisyFiltered"] (make_golden_gicp_ndt.py): the scan against a copy moved by
    case["P"] and re-measured with Gaussian noise of a fixed seed -- regenerated here, checked against
    the checksum the generator recorded."""
    P = np.array(case["P"])
    rng = np.random.Generator(np.random.PCG64(case["noise_seed"]))
    moved = (scan.astype(np.float64) @ P[:3, :3].T + P[:3, 3]).astype(np.float32)
    noisy = (moved.astype(np.float64) + rng.normal(0.0, case["noise_sigma"], scan.shape)).astype(np.float32)
    assert abs(float(np.abs(noisy.astype(np.float64)).sum()) - case["target_checksum"]) <= 1e-9 * case["target_checksum"]
    return noisy, P
