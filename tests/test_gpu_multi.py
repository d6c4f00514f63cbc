"""N > 1 on hardware: one process per GPU under torch.distributed.run, RCCL over xGMI (band reduce vs host sum, all-gather
exchange of the band pool, band-sharded primal-dual loop vs the single-process loop, row-sharded gridder vs the unsharded
plan).  The all-GPU case needs >= 2 visible GPUs; the single-rank case runs the same script on any GPU box."""

import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ngpu():
    from pfb_imaging_amd import _lib

    return _lib.device_count()


def _launch(nproc):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    # fresh child processes: the launcher starts before anything in them touches a GPU
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_rccl_worker.py")]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-6000:]
    for r in range(nproc):
        assert f"rank {r} of {nproc} ok" in p.stdout, p.stdout[-3000:]


def test_rccl_worker_single_rank():
    _launch(1)


@pytest.mark.skipif(_ngpu() < 2, reason="needs at least two GPUs")
def test_rccl_worker_one_rank_per_gpu():
    _launch(min(_ngpu(), 8))
