"""What the sharded path's mailboxes rest on, between two PROCESSES on one GPU (tests/cpp_ipc/ipc_mailbox.hip):
uncached device memory exported / mapped through HIP IPC handles, system-scope stores of the importing process
seen by a kernel of the owning process that is already running and polling.  (Between GPUs the same calls go over
xGMI; that has not run in the builder's container.)"""
import os
import subprocess
import tempfile

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cpp_ipc", "ipc_mailbox.hip")
BIN = os.path.join(HERE, "cpp_ipc", "ipc_mailbox")


def _binary():
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < os.path.getmtime(SRC):
        hipcc = "/opt/rocm/bin/hipcc"
        if not os.path.exists(hipcc):
            pytest.skip("no hipcc to build the probe")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-o", BIN, SRC])
    return BIN


def test_polling_kernel_sees_another_process_stores_through_an_ipc_mapping():
    exe = _binary()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    with tempfile.TemporaryDirectory() as d:
        base = os.path.join(d, "box")
        owner = subprocess.Popen([exe, "owner", base], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        peer = subprocess.Popen([exe, "peer", base], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        po, pe = peer.communicate(timeout=120)
        oo, oe = owner.communicate(timeout=120)
    assert peer.returncode == 0, (po, pe)
    assert owner.returncode == 0 and "saw every word" in oo, (oo, oe)
