"""The optimisers' scalar code is one source for the host and the device (libwave_amd/csrc/wm_bfgs.hpp: pcl::BFGS with
Fletcher's line search; wm_ndt_ctl.hpp: Eigen's JacobiSVD solve, More-Thuente).  Here it is compiled with g++ and run
away from any device: BFGS on a 6-D bowl whose minimum is known (restarted per outer iteration as GICP restarts it),
the 6 x 6 SVD solve on a regular and on a rank-deficient system."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_shared_control_code_on_the_host(tmp_path):
    exe = str(tmp_path / "ctl_host")
    build = subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I" + os.path.join(ROOT, "libwave_amd", "csrc"),
                            os.path.join(ROOT, "tests", "cpp_host", "ctl_host.cpp"), "-o", exe, "-lm"],
                           capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "failed checks: 0" in run.stdout, run.stdout + run.stderr[-1000:]
