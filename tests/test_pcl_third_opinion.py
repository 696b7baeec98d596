"""Optional third opinion from a REAL PCL: runs only where oracle/pcl_ref built
its PCL-backed tool (a system PCL >= 1.8; not present in the build image -- then these tests skip).
The reference's smallDisplacement cases on its own fixture: the C oracle must agree with PCL's final
transform to north_star's 1e-4 m / 1e-4 rad."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import golden_checks as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "oracle", "_ref", "pcl_third_opinion")

pytestmark = pytest.mark.skipif(
    not os.path.exists(TOOL),
    reason="no system PCL: oracle/pcl_ref not built -- parity stays UNPINNED by the reference.  The one command "
           "that flips it to pinned, on a box with network: `sudo apt-get install -y libpcl-dev pkg-config && "
           "make -C oracle/pcl_ref && python -m pytest tests/test_pcl_third_opinion.py -q` (oracle/pcl_ref/README.md)")


def _pcl(kind, ref, tgt, res=None):
    with tempfile.TemporaryDirectory() as d:
        a, b = os.path.join(d, "ref.f32"), os.path.join(d, "tgt.f32")
        np.ascontiguousarray(ref, np.float32).tofile(a)
        np.ascontiguousarray(tgt, np.float32).tofile(b)
        out = subprocess.run([TOOL, kind, a, b] + ([str(res)] if res else []), capture_output=True, text=True,
                             timeout=1800).stdout.split()
    return int(out[0]), np.array([float(v) for v in out[1:17]]).reshape(4, 4)


@pytest.mark.parametrize("kind,res", [("icp", None), ("gicp", None), ("ndt", 0.3)])
def test_oracle_agrees_with_real_pcl(oracle, testscan, kind, res):
    target, P = G.shifted(testscan, 0.2)
    ok, T = _pcl(kind, testscan, target, res)
    assert ok and np.linalg.norm(T - P) < 0.12
    if kind == "icp":
        want = oracle.icp_align(testscan, target)["T"]
    elif kind == "gicp":
        want = oracle.gicp_align(testscan, target)["T"]
    else:
        want = oracle.ndt_align(testscan, target, res=res, step_size=3, max_iter=100, t_eps=1e-8)["T"]
    dt, ang = G.pose_err(T, want)
    assert dt <= 1e-4 and ang <= 1e-4, (kind, dt, ang)
