"""world_size-2 (and 3) gloo runs of the multi-GPU orchestration on CPU."""
import os
import subprocess
import sys

import pytest

from tests.paf_gen import random_paf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_world(tmp_path, world, backend, port):
    text, _ = random_paf(77, 260, n_seq=7, seq_len=20000, self_aln=True)
    path = str(tmp_path / "w.paf")
    with open(path, "w") as f:
        f.write(text)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tests", "sharded_worker.py"), backend, path],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "sharded ok world=%d" % world in r.stdout


@pytest.mark.parametrize("world", [1, 2, 3])
def test_sharded_gloo(tmp_path, world):
    run_world(tmp_path, world, "cpu", 29600 + world)


@pytest.mark.gpu
def test_sharded_gpu_world1(tmp_path):
    """The same orchestration on the HIP stage API with RCCL (one rank: the GPU box has one GPU)."""
    run_world(tmp_path, 1, "gpu", 29650)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_gpu_multirank_over_gloo(tmp_path, world):
    """Several ranks, each with its own shard of the index on the (single) GPU and the
    real stage API; only the transport differs from the RCCL run (host memory)."""
    run_world(tmp_path, world, "gpu-gloo", 29660 + world)
