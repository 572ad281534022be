"""Multi-GPU pieces that need no GPU: the shard map, and the host transport of the sharded engine driven
through the C ABI by 2 and 3 gloo ranks (impg_gpu_comm_create_host + impg_gpu_comm_check: the all-gather
and the ragged all-to-all-v every hop of a sharded query is made of)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import impg_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_ranks(world, args, port, lanes=2, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", IMPG_TEST_LANES=str(lanes))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tests", "multi_worker.py")] + args,
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("world", [1, 2, 3])
def test_host_transport_gloo(world):
    out = run_ranks(world, ["comm-check", "host"], 29700 + world)
    assert "comm ok world=%d lanes=2" % world in out


def test_shard_assign_balances_and_is_deterministic():
    rng = np.random.default_rng(5)
    for n_seq, n_shards in ((1, 1), (7, 3), (200, 8), (20000, 8), (5, 8)):
        cnt = rng.integers(0, 50000, n_seq).astype(np.uint64)
        cnt[rng.random(n_seq) < 0.1] = 0
        if n_seq >= 200:
            cnt[:3] = [4_000_000, 2_500_000, 900_000]  # a few chromosomes dominate, as on real assemblies
        own = impg_amd.shard_assign(cnt, n_shards)
        assert own.tolist() == impg_amd.shard_assign(cnt, n_shards).tolist()
        assert own.max(initial=0) < n_shards
        load = np.bincount(own, weights=cnt.astype(np.float64), minlength=n_shards)
        # greedy longest-first: no shard exceeds the mean by more than the largest item
        assert load.max() <= cnt.sum() / n_shards + cnt.max(initial=0) + 1
        if n_seq >= 200:  # and far better than target_id % n_shards on skewed data
            naive = np.bincount(np.arange(n_seq) % n_shards, weights=cnt.astype(np.float64), minlength=n_shards)
            assert load.max() <= naive.max()
    with pytest.raises(impg_amd.ImpgGpuError):
        impg_amd.shard_assign([1, 2, 3], 0)
