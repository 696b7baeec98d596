"""The cases of the reference's gtest suites (wave_matching/tests/{icp,gicp,ndt,multi_matcher}
_tests.cpp) run in C++ against the drop-in wave:: API: tests/cpp/matcher_cases.cpp (a table of
registrations + a few construction / estimator / pool cases), built by libwave_amd/host/Makefile."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "wave_matching_tests")


def _run(filt=None, timeout=600):
    import __graft_entry__ as g
    g.build()
    env = dict(os.environ, WAVE_TEST_ROOT=ROOT)
    cmd = [BIN] + ([filt] if filt else [])
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_cpp_shim_builds_and_links():
    import __graft_entry__ as g
    g.build()
    assert os.path.exists(BIN)
    out = subprocess.run(["nm", "-D", "--defined-only",
                          os.path.join(ROOT, "libwave_amd", "libwave_matching.so")],
                         capture_output=True, text=True).stdout
    for sym in ("ICPMatcher5matchEv", "ICPMatcher12estimateInfoEv", "ICPMatcherParamsC1ERKNSt"):
        assert sym in out, sym


def test_cpp_yaml_params_on_cpu():
    r = _run("paramsFromYaml")
    assert r.returncode == 0, r.stdout + r.stderr


def test_cpp_construction_cases_on_cpu():
    """Matcher / MultiMatcher construction and teardown need no device (contexts are created at
    the first match): the four *.initialization cases, incl. the worker pool's start and join."""
    r = _run("initialization")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "4 tests ran, 0 failed" in r.stdout


@pytest.mark.gpu
def test_reference_gtest_cases_pass_on_gpu():
    r = _run()
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 failed" in r.stdout
