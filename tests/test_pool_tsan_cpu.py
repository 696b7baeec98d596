"""ThreadSanitizer run of wave::MultiMatcher's worker pool (tests/cpp_tsan/pool_tsan.cpp): two
producers, six workers and a consumer hammer the queues; any data race TSan sees fails the test.
CPU only -- the pool is a template, exercised here with a matcher that needs no device."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_multimatcher_pool_is_race_free_under_tsan(tmp_path):
    exe = str(tmp_path / "pool_tsan")
    build = subprocess.run(["g++", "-std=c++14", "-O1", "-g", "-fsanitize=thread", "-I" + os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "tests", "cpp_tsan", "pool_tsan.cpp"), "-o", exe, "-lpthread"],
                           capture_output=True, text=True, timeout=600)
    if build.returncode != 0 and ("libtsan" in build.stderr or "-ltsan" in build.stderr):
        pytest.skip("no ThreadSanitizer runtime in this toolchain")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66"))
    assert "WARNING: ThreadSanitizer" not in run.stderr, run.stderr[-3000:]
    assert run.returncode == 0 and "OK" in run.stdout, (run.returncode, run.stdout, run.stderr[-1000:])
    # the reference's default queue of 10 must not cap the batches (they are gathered over several refills)
    assert "queue of 10: largest batch" in run.stdout and "capped by the queue" not in run.stdout, run.stdout
    # getResult() blocks while pairs are pending and returns false once all have been handed out
    # (wave_matching/include/wave/matching/multi_matcher.hpp:64-77)
    assert "blocking getResult: 700 of 700 collected, 6 late" in run.stdout, run.stdout
    # every device slot of an eight-device list is fed batches
    assert "device slots: batches per slot" in run.stdout and "got no batches" not in run.stdout, run.stdout
