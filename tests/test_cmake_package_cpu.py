"""The CMake face of the drop-in (reference: find_package(wave ... matching) + wave::matching,
README.md:100-103, wave_matching/CMakeLists.txt:3-15): configure, build and install this repository
with CMake (hipcc cross-compiles gfx950 without a GPU), then configure and build a consumer project
against the installed package.  Build-only: nothing here needs a device."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCXX = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(shutil.which("cmake") is None or shutil.which("ninja") is None or not os.path.exists(HIPCXX),
                    reason="needs cmake, ninja and the ROCm clang")
def test_cmake_package_builds_installs_and_is_found(tmp_path):
    build, inst, cons = tmp_path / "build", tmp_path / "inst", tmp_path / "consumer"
    run = lambda *a, **k: subprocess.run(*a, check=True, capture_output=True, text=True, timeout=1500, **k)  # noqa: E731
    run(["cmake", "-G", "Ninja", "-S", ROOT, "-B", str(build), "-DCMAKE_HIP_COMPILER=" + HIPCXX,
         "-DCMAKE_INSTALL_PREFIX=" + str(inst), "-DWAVE_BUILD_TESTS=OFF"])
    run(["cmake", "--build", str(build), "--target", "install", "-j", "8"])
    assert (inst / "lib" / "libwave_matching.so").exists() and (inst / "lib" / "libwavematch_hip.so").exists()
    assert (inst / "include" / "wave" / "matching" / "icp.hpp").exists()
    run(["cmake", "-G", "Ninja", "-S", os.path.join(ROOT, "tests", "cmake_consumer"), "-B", str(cons),
         "-DCMAKE_PREFIX_PATH=" + str(inst)])
    run(["cmake", "--build", str(cons)])
    assert (cons / "consumer").exists()
