#!/usr/bin/env python3
"""Differential soak test of the multi-GPU orchestration: W ranks (sharing the box's one GPU, collectives over
gloo) each with its shard of a random index and its own random queries, against the CPU oracle.
usage: fuzz_sharded.py <seconds> [world] [first_seed]      (re-executes itself under torch.distributed.run)"""
import os, sys, time, tempfile, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if "RANK" not in os.environ:
    secs = sys.argv[1] if len(sys.argv) > 1 else "60"
    world = sys.argv[2] if len(sys.argv) > 2 else "2"
    seed = sys.argv[3] if len(sys.argv) > 3 else "5000"
    if not 1 <= int(world) <= 8:
        sys.exit("world must be 1..8 (usage: fuzz_sharded.py <seconds> [world] [first_seed])")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", world, "--master-addr",
                        "127.0.0.1", "--master-port", "29701", os.path.abspath(__file__), secs, world, seed], cwd=ROOT)
    sys.exit(r.returncode)

import numpy as np, torch, torch.distributed as dist
import impg_amd
from impg_amd.sharded import GpuBackend, ShardedImpg
from oracle import oracle as o
from tests.paf_gen import random_paf, random_ranges

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
budget, seed = float(sys.argv[1]), int(sys.argv[3])
t_end = time.time() + budget
tmp = tempfile.gettempdir()
n_cases = 0
while True:
    go = torch.tensor([1 if time.time() < t_end else 0])
    dist.broadcast(go, 0)
    if not int(go):
        break
    rng = np.random.default_rng(seed)  # same stream on every rank
    n_seq = int(rng.integers(2, 14)); seq_len = int(rng.choice([2500, 8000, 30000])); max_ops = int(rng.choice([6, 30, 150]))
    n_rec = int(rng.integers(20, 500)); weird = bool(rng.random() < 0.4); self_aln = bool(rng.random() < 0.6)
    path = os.path.join(tmp, "fs_%d.paf" % seed)
    if rank == 0:
        text, _ = random_paf(seed * 3, n_rec, n_seq=n_seq, seq_len=seq_len, max_ops=max_ops, weird=weird, self_aln=self_aln)
        open(path, "w").write(text)
    dist.barrier()
    c = o.OracleIndex(paf_paths=[path], preparse=True)
    g = impg_amd.GpuImpg.from_paf(path, device=0, shard=rank, n_shards=world)
    g.set_option("locality_min", int(rng.choice([0, 1, 4096])))
    eng = ShardedImpg(GpuBackend(g, 0), rank, world, torch.device("cuda", 0), chunk_ranges=int(rng.choice([3, 11, 1000])))
    eng.A2A_ROUND_BYTES = int(rng.choice([120, 4096, 512 << 20]))
    kw = {}
    if rng.random() < 0.75:
        kw.update(transitive=True, max_depth=int(rng.choice([1, 2, 3])), min_transitive_len=int(rng.choice([0, 10, 101])),
                  min_distance_between_ranges=int(rng.choice([0, 10, 200])))
    if rng.random() < 0.3:
        kw["min_output_length"] = int(rng.choice([0, 100, 1000]))
    if rng.random() < 0.3:
        kw["min_identity"] = float(rng.choice([0.3, 0.7]))
    rl = random_ranges(seed * 11 + rank, int(rng.integers(1, 40)) + 3 * rank, c.num_seqs(), seq_len, max_len=min(3000, seq_len - 1), min_len=1)
    ranges = np.array(rl, dtype=impg_amd.RANGE_DTYPE)
    p = impg_amd.make_params(**kw)
    got = eng.query_batch(ranges, p)
    total = 0
    for i, (t, s, e) in enumerate(rl):
        want = c.query(t, s, e, **kw)
        assert got[i].tolist() == want.tolist(), ("rows", seed, rank, i, kw)
        total += c.last_projection_count()
    rt = torch.from_numpy(ranges.view(np.uint8).copy()).to(eng.device)
    st = eng.query_batch_stats(rt, len(rl), p)
    tt = torch.tensor([st.projected, total], dtype=torch.int64)
    dist.all_reduce(tt)
    assert int(tt[0]) == int(tt[1]), ("projected", seed, rank, kw, tt.tolist())
    dist.barrier()
    if rank == 0:
        os.remove(path)
    n_cases += 1
    seed += 1
if rank == 0:
    print("sharded fuzz ok: world %d, %d cases, seeds up to %d" % (world, n_cases, seed - 1))
dist.destroy_process_group()
