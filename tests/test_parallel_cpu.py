"""CPU tests of the band-parallel runtime: sharding map, and the N > 1 path with world_size 2 -- over the product's own
TCP rendezvous / host transport (plain processes, no torch), and over gloo (torch.distributed.run on 127.0.0.1) standing in
for it.  The host transport stands in for RCCL where there is no GPU."""

import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_band_sharding_map():
    from pfb_imaging_amd.parallel import band_owner, local_bands

    assert [band_owner(b, 8) for b in range(8)] == list(range(8))
    assert local_bands(8, 3, 4) == [3, 7]
    assert local_bands(3, 5, 8) == []
    owners = [band_owner(b, 3) for b in range(10)]
    assert sorted(sum((local_bands(10, r, 3) for r in range(3)), [])) == list(range(10))
    assert owners.count(0) - owners.count(2) <= 1


def test_single_process_pool_is_local():
    from pfb_imaging_amd.operators.band_worker import BandWorkerPool
    from pfb_imaging_amd.parallel import BandComm

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _gloo_worker import FakeWorker

    comm = BandComm(0, 1, 0, transport="host")
    assert np.array_equal(comm.reduce_sum(np.arange(3.0)), np.arange(3.0))
    assert comm.max_over_ranks(2.0) == 2.0
    pool = BandWorkerPool(3, comm=comm, worker_cls=FakeWorker)
    assert pool.local == [0, 1, 2]
    d = [np.full((2, 2), 1.0 + b) for b in range(3)]
    pool.init_hess(d, 2, 2, 4, 4, np.zeros(3), [None] * 3)
    x = np.ones((3, 2, 2))
    np.testing.assert_array_equal(pool.hess_dot(x), np.stack(d))


def test_two_rank_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_gloo_worker.py"), "gloo"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "rank 0 ok" in p.stdout and "rank 1 ok" in p.stdout


def test_two_rank_socket_bootstrap():
    """The product's rendezvous and host collectives: two plain processes, RANK / WORLD_SIZE / MASTER_* in the environment,
    a foreign listener squatting on the first candidate port (MASTER_PORT + 1) that must be skipped."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    squat = socket.socket()
    try:
        squat.bind(("127.0.0.1", port + 1))
        squat.listen(4)  # accepts (kernel backlog) and never answers the hello
    except OSError:
        squat.close()
        squat = None
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                   WORLD_SIZE="2", OMP_NUM_THREADS="1", PFBHIP_RDZV_TIMEOUT="120")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gloo_worker.py"), "socket"], env=env,
                                      cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=280) for p in procs]
    if squat is not None:
        squat.close()
    for rank, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, so[-2000:] + se[-4000:]
        assert f"rank {rank} ok" in so


def test_two_rank_socket_under_the_launcher():
    """The way the driver starts bench.py for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P ...` -- the launcher's own store owns MASTER_PORT; the product's rendezvous must come up beside it
    from the environment the launcher sets (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT), without importing torch."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1", PFBHIP_RDZV_TIMEOUT="120")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_gloo_worker.py"), "socket"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "rank 0 ok" in p.stdout and "rank 1 ok" in p.stdout
