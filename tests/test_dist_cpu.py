"""The N>1 path of bench.py on CPU: world_size 2 over gloo (one process per rank, as the driver launches it)."""
import os
import socket
import subprocess
import sys

from conftest import ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_barrier_max_and_sharding(tmp_path):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "_dist_worker.py"), str(tmp_path)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()
    assert (tmp_path / "rank0.ok").read_text() == (tmp_path / "rank1.ok").read_text()  # both ranks see the same maxima
