"""Worker of the multi-rank tests: every rank shards the index by target,
submits its own queries, and checks its results against the oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import impg_amd  # noqa: E402
from impg_amd.sharded import GpuBackend, ShardedImpg  # noqa: E402
from oracle import oracle as o  # noqa: E402
from tests.paf_gen import random_paf, random_ranges  # noqa: E402


def main():
    # "cpu": gloo + oracle stand-in; "gpu": nccl (RCCL) + HIP engine; "gpu-gloo": HIP engine on
    # every rank (ranks share the one GPU of the test box), collectives through host memory
    backend_name = sys.argv[1]
    paf_path = sys.argv[2]
    dist.init_process_group("nccl" if backend_name == "gpu" else "gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    c = o.OracleIndex(paf_paths=[paf_path], preparse=True)
    n_seq = c.num_seqs()
    if backend_name == "cpu":
        from tests.cpu_backend import OracleBackend
        eng = ShardedImpg(OracleBackend(c, rank, world), rank, world, torch.device("cpu"), chunk_ranges=7)
    else:
        torch.cuda.set_device(0)
        g = impg_amd.GpuImpg.from_paf(paf_path, device=0, shard=rank, n_shards=world)
        eng = ShardedImpg(GpuBackend(g, 0), rank, world, torch.device("cuda", 0), chunk_ranges=7)
        # impg_gpu_stage_route: stable partition by target_id % W, qidx := home index, counts per owner
        gen = torch.Generator(device="cpu").manual_seed(3 + rank)
        for n in (1, 777, 300_000):
            for W in (1, 2, 3, 8):
                fr = torch.randint(0, 1000, (n, 4), dtype=torch.int32, generator=gen).cuda()
                out = torch.empty_like(fr)
                torch.cuda.synchronize()
                counts = g.stage_route(fr.data_ptr(), n, W, out.data_ptr())
                owner = torch.remainder(fr[:, 0].to(torch.int64), W)
                order = torch.argsort(owner, stable=True)
                want = fr[order].clone()
                want[:, 3] = order.to(torch.int32)
                assert bool((out == want).all()), (n, W)
                assert [int(x) for x in counts] == torch.bincount(owner, minlength=W).tolist()
        # impg_gpu_stage_reorder: blocks (one per owner) of ascending fidx runs -> the stable order by fidx
        for cols in (4, 8):
            for n_front, W in ((1, 1), (50, 3), (200_000, 8)):
                owner = torch.randint(0, W, (n_front,), generator=gen)
                reps = torch.randint(0, 40, (n_front,), generator=gen)
                reps[torch.rand(n_front, generator=gen) < 0.01] = 3000  # a few long runs
                blocks = []
                for w in range(W):
                    idx = torch.nonzero(owner == w).flatten()
                    blocks.append(torch.repeat_interleave(idx, reps[idx]))
                fidx = torch.cat(blocks).to(torch.int32)
                hits = torch.randint(-5, 1 << 30, (fidx.numel(), cols), dtype=torch.int32, generator=gen)
                hits[:, 0] = fidx
                hits = hits.cuda()
                want = hits[torch.argsort(hits[:, 0].to(torch.int64), stable=True)]
                got = eng.backend.reorder(hits, n_front)
                assert got.shape == want.shape and bool((got == want).all()), (cols, n_front, W)
        bad = torch.tensor([[0, 1, 2, 3], [1, 1, 2, 3], [0, 1, 2, 3]], dtype=torch.int32).cuda()  # fidx 0 in two runs
        try:
            eng.backend.reorder(bad, 2)
            raise AssertionError("a split run must be rejected")
        except impg_amd.ImpgGpuError:
            pass
        eng.always_reorder = True  # one rank still takes the home-side reorder below
    seq_len = int(c.seq_len(0))
    n_q = 18 + 5 * rank  # ranks own different numbers of queries
    rl = random_ranges(100 + rank, n_q, n_seq, seq_len, max_len=3000, min_len=120)
    ranges = np.array(rl, dtype=impg_amd.RANGE_DTYPE)
    cases = [dict(), dict(transitive=True, max_depth=2), dict(transitive=True, max_depth=3, min_transitive_len=20),
             dict(transitive=True, max_depth=0, min_transitive_len=200, min_output_length=150)]
    for ci, kw in enumerate(cases):
        # odd cases: exchanges cut into many small rounds (the > 512 MB path of _all_to_all_rows)
        eng.A2A_ROUND_BYTES = (512 << 20) if ci % 2 == 0 else 200
        p = impg_amd.make_params(**kw)
        got = eng.query_batch(ranges, p)
        total = 0
        for i, (t, s, e) in enumerate(rl):
            want = c.query(t, s, e, **kw)
            assert got[i].tolist() == want.tolist(), (rank, i, kw)
            total += c.last_projection_count()
        rt = torch.from_numpy(ranges.view(np.uint8).copy()).to(eng.device)
        st = eng.query_batch_stats(rt, n_q, p)
        # projections are counted where they are computed: compare the global sums
        tt = torch.tensor([st.projected, total], dtype=torch.int64, device=eng.comm_device)
        dist.all_reduce(tt)
        assert int(tt[0]) == int(tt[1]), (rank, kw, tt.tolist())
    dist.barrier()
    if rank == 0:
        print("sharded ok world=%d" % world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
