"""bench.py's counter-traffic plumbing (no GPU): the committed PMC summary carries FETCH / WRITE for the
kernels the line quotes, and the helper turns them into bytes per launch and a share of the copy rate."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_committed_pmc_summary_covers_the_quoted_kernels():
    import json
    with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
        pmc = json.load(f)
    assert pmc.get("tag"), "profiles/pmc_latest.json missing or without a tag"
    assert len(pmc.get("csrc_sha256", "")) == 64, "capture not stamped with the kernel sources' hash"
    for k in ("k_nn_grid", "k_nn_cert", "k_gicp_fdf", "k_gicp_quad", "k_ndt_derivs"):
        assert "FETCH_SIZE_kb_per_dispatch" in pmc.get(k, {}), k
        assert "WRITE_SIZE_kb_per_dispatch" in pmc.get(k, {}), k
    assert pmc["k_ndt_derivs"].get("f64_flops_per_source_point", 0) > 100


def test_counter_traffic_arithmetic():
    import bench
    pmc = {"tag": "t", "k": {"FETCH_SIZE_kb_per_dispatch": 1000.0, "WRITE_SIZE_kb_per_dispatch": 500.0}}
    out = bench.counter_traffic(pmc, "k", avg_launch_us=10.0, copy_peak_gbs=5000.0)
    assert out["traffic"] == (2 * 1000.0 + 500.0) * 1024.0          # FETCH doubled on gfx950, WRITE as reported
    assert abs(out["hbm_util"] - out["traffic"] / 10e-6 / 1e9 / 5000.0) < 1e-12
    assert bench.counter_traffic(pmc, "absent", 10.0, 5000.0) == {"traffic": None}


def test_stale_capture_reports_no_traffic(tmp_path, monkeypatch):
    """A capture stamped with other kernel sources than the tree's must not be quoted."""
    import json
    import bench
    fake = tmp_path / "profiles"
    fake.mkdir()
    (fake / "pmc_latest.json").write_text(json.dumps({
        "tag": "old", "csrc_sha256": "0" * 64,
        "k_nn_grid": {"FETCH_SIZE_kb_per_dispatch": 1.0, "WRITE_SIZE_kb_per_dispatch": 1.0}}))
    real_root = bench.ROOT
    real_hash = bench.csrc_sha256()
    monkeypatch.setattr(bench, "csrc_sha256", lambda: real_hash)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    stale = bench.pmc_summary()
    assert "stale" in stale and "tag" not in stale
    assert bench.pmc_traffic_bytes(stale, "k_nn_grid") is None
    assert bench.counter_traffic(stale, "k_nn_grid", 10.0, 5000.0) == {"traffic": None}
    (fake / "pmc_latest.json").write_text(json.dumps({
        "tag": "new", "csrc_sha256": real_hash,
        "k_nn_grid": {"FETCH_SIZE_kb_per_dispatch": 1.0, "WRITE_SIZE_kb_per_dispatch": 1.0}}))
    assert bench.pmc_summary().get("tag") == "new"
    # a capture WITHOUT a stamp cannot be tied to the kernels in the tree: stale as well
    (fake / "pmc_latest.json").write_text(json.dumps({
        "tag": "unstamped", "k_nn_grid": {"FETCH_SIZE_kb_per_dispatch": 1.0, "WRITE_SIZE_kb_per_dispatch": 1.0}}))
    unstamped = bench.pmc_summary()
    assert "stale" in unstamped and "tag" not in unstamped
    monkeypatch.setattr(bench, "ROOT", real_root)


def test_rotation_angle_of_identical_float_matrices_is_tiny():
    """The parity figure of the bench line: two identical float rotations must read ~1e-7 rad, not the 3e-4 that
    arccos((trace - 1) / 2) makes of their non-orthonormality."""
    import numpy as np
    import bench
    c, s_ = np.float32(np.cos(0.3)), np.float32(np.sin(0.3))
    R = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]], dtype=np.float32).astype(np.float64)
    assert bench.rotation_angle(R, R) < 1e-6
    R2 = np.array([[np.cos(0.3001), -np.sin(0.3001), 0], [np.sin(0.3001), np.cos(0.3001), 0], [0, 0, 1]])
    assert abs(bench.rotation_angle(R, R2) - 1e-4) < 2e-6


def test_calibrated_gather_traffic():
    """A kernel that streams S bytes and gathers the rest: FETCH_SIZE counts the streams at half (calibration:
    profiles/r04_fetch_size_calibration.json), so bytes = FETCH + S / 2 + WRITE; never below the streams."""
    import bench
    pmc = {"k": {"FETCH_SIZE_kb_per_dispatch": 1000.0, "WRITE_SIZE_kb_per_dispatch": 100.0}}
    assert bench.pmc_traffic_bytes(pmc, "k") == (2000.0 + 100.0) * 1024.0
    assert bench.pmc_traffic_bytes(pmc, "k", stream_bytes=400.0 * 1024.0) == (1000.0 + 200.0 + 100.0) * 1024.0
    assert bench.pmc_traffic_bytes(pmc, "k", stream_bytes=4000.0 * 1024.0) == (4000.0 + 100.0) * 1024.0
    import json
    cal = json.load(open(os.path.join(ROOT, "profiles", "r04_fetch_size_calibration.json")))
    assert abs(cal[0]["factor_to_64B_sector_bytes"] - 2.0) < 0.05        # the stream: counted at half
    assert abs(cal[3]["factor_to_128B_line_bytes"] - 1.0) < 0.05         # four float4 of one line: the whole line, once


def test_multi_gpu_bench_line_assembles_from_the_sharded_result_keys():
    """`bench.py --gpus N` has never run on more than one physical GPU in the build container: what CAN be checked
    without GPUs is that rank 0's line is assembled from exactly the keys the C-ABI wrapper produces for a sharded
    registration (capi.Context._sharded_dict over wm_icp_stats) -- no KeyError on the first real 8-GPU run, the
    contract's fields present, `config.sharding` carrying the per-phase budget, JSON-serialisable."""
    import argparse
    import json
    import numpy as np
    import bench
    from libwave_amd import capi
    st = capi.IcpStats()
    st.iterations, st.n_corr, st.nn_ms, st.nn_launches, st.cert_launches, st.nn_cert_ms = 50, 8_000_000, 3.1, 50, 25, 0.7
    st.plan_ms, st.compact_ms, st.index_ms, st.iter_ms, st.allreduce_ms = 0.4, 0.2, 0.5, 3.6, 0.6
    st.n_tgt_local, st.n_src_local, st.rccl_ranks, st.shard_attempts, st.grid_cell = 1_050_000, 1_020_000, 8, 1, 0.183
    r = capi.Context._sharded_dict(0, np.eye(4), st)
    a = argparse.Namespace(points=1_000_000, steps=20, warmup=5, iters=50, max_corr=3.0)
    for world in (2, 8):
        out = bench.assemble_line(a, world, r, np.eye(4), elapsed=0.1, step_ms=[5.0] * 20, nn_ms=3.1, nn_launches=50,
                                  cert_ms=0.7, cert_launches=25, parallelism="target x-slabs x%d" % world,
                                  sharded=True, ar_us=11.0)
        line = json.loads(json.dumps(out))
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in line, k
        assert line["n_gpus"] == world and line["scaling"] == "weak" and line["vs_baseline"] is None
        # whole-job throughput in 1M-point registration equivalents: N x the raw rate
        assert abs(line["value"] - world * 20 / 0.1) < 1e-6
        sh = line["config"]["sharding"]
        for k in ("plan_ms", "compact_ms", "index_ms", "iter_ms", "allreduce_ms", "n_tgt_local", "n_src_local",
                  "rccl_ranks", "shard_attempts", "owned_violations", "allreduce_us_isolated"):
            assert k in sh, k
        assert sh["rccl_ranks"] == 8 and sh["allreduce_us_isolated"] == 11.0
        assert line["roofline"]["traffic"] is None and line["roofline"]["frac"] > 0
    # ... and the one-GPU line (no sharding block)
    one = bench.assemble_line(a, 1, capi.Context._stats_dict(0, np.eye(4), st), np.eye(4), 0.075, [3.75] * 20, 3.1, 50, 0.7, 25,
                              "single", pmc={}, peak_copy=5800.0)
    assert "sharding" not in one["config"] and one["n_gpus"] == 1 and json.dumps(one)


def test_first_sharded_step_falls_back_once_and_only_together():
    """bench.py --gpus N: the first sharded registration decides, for all ranks alike, whether the run keeps the
    mailbox exchange; a failure anywhere rebuilds everywhere, a second failure is an error."""
    import bench

    class Boom(RuntimeError):
        pass
    calls = {"step": 0, "rebuild": 0, "votes": []}

    def votes(ok):  # (one rank here: the AND over the ranks is its own vote; a peer's failure is injected below)
        calls["votes"].append(ok)
        return ok

    def good():
        calls["step"] += 1
    assert bench.first_step_with_fallback(good, lambda: calls.__setitem__("rebuild", calls["rebuild"] + 1), votes, Boom) is False
    assert calls == {"step": 1, "rebuild": 0, "votes": [True]}

    # this rank succeeds, a peer does not: this rank rebuilds too and registers again
    calls.update(step=0, rebuild=0, votes=[])
    peer = iter([False, True])
    assert bench.first_step_with_fallback(good, lambda: calls.__setitem__("rebuild", calls["rebuild"] + 1),
                                          lambda ok: votes(ok) and next(peer), Boom) is True
    assert calls["step"] == 2 and calls["rebuild"] == 1

    # this rank fails once (the mailbox exchange timed out), then the collective exchange works
    state = {"n": 0}

    def flaky():
        state["n"] += 1
        if state["n"] == 1:
            raise Boom("a peer's block did not arrive")
    calls.update(step=0, rebuild=0, votes=[])
    assert bench.first_step_with_fallback(flaky, lambda: calls.__setitem__("rebuild", calls["rebuild"] + 1), votes, Boom) is True
    assert calls["rebuild"] == 1 and calls["votes"] == [False, True]

    # failing twice is fatal
    def bad():
        raise Boom("no")
    with pytest.raises(RuntimeError):
        bench.first_step_with_fallback(bad, lambda: None, votes, Boom)
