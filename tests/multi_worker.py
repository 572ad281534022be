"""Worker of the multi-rank tests (one process per rank, launched by torch.distributed.run).

  comm-check <transport>   CPU: the host transport (gloo callbacks) passes impg_gpu_comm_check on every lane
  fail <transport> <paf>   GPU: a rank fails in the middle of a batch (injected): every rank gets an error, none hangs
  query <transport> <paf>  GPU: every rank builds its shard, submits its own queries (collective calls) and
                           checks its results against the oracle; transport = host (gloo, ranks share GPU 0)
                           or rccl (one GPU per rank)
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import impg_amd  # noqa: E402


def make_comm(transport, rank, world, lanes, device):
    if transport == "rccl":
        return impg_amd.Comm.rccl(rank, world, device, lanes=lanes)
    groups = [dist.new_group(list(range(world)), backend="gloo") for _ in range(lanes)]
    return impg_amd.Comm.host(rank, world, device, groups)


def main():
    mode, transport = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    lanes = int(os.environ.get("IMPG_TEST_LANES", "2"))
    if mode == "comm-check":
        comm = make_comm(transport, rank, world, lanes, 0)
        assert comm.kind() == "host"
        comm.check()
        dist.barrier()
        if rank == 0:
            print("comm ok world=%d lanes=%d" % (world, lanes))
        comm.close()
        dist.destroy_process_group()
        return

    from oracle import oracle as o
    from tests.paf_gen import random_ranges
    paf_path = sys.argv[3]
    device = int(os.environ.get("LOCAL_RANK", "0")) if transport == "rccl" else 0
    comm = make_comm(transport, rank, world, lanes, device)
    comm.check()
    c = o.OracleIndex(paf_paths=[paf_path], preparse=True)
    g = impg_amd.GpuImpg.from_paf(paf_path, device=device, comm=comm)
    r, w, l, owner = g.shard_info()
    assert (r, w, l) == (rank, world, lanes)
    g.set_option("chunk_ranges", 7)
    n_seq, seq_len = c.num_seqs(), int(c.seq_len(0))
    if mode == "fail":  # failure agreement: rank 0 fails on the owner side of hop 2, then rank `world - 1` on the home side
        rl = random_ranges(100 + rank, 20, n_seq, seq_len, max_len=3000, min_len=120)
        kw = dict(transitive=True, max_depth=3, min_transitive_len=20)
        for side, who in (("debug_fail_owner", 0), ("debug_fail_home", world - 1)):
            g.set_option(side, (who + 1) << 16 | 2)
            try:
                g.query_batch(rl, impg_amd.make_params(**kw))
                raise AssertionError("rank %d: the batch must fail" % rank)
            except impg_amd.ImpgGpuError as e:
                assert ("injected failure" in str(e)) == (rank == who), (rank, side, str(e))
            g.set_option(side, 0)
            got = g.query_batch(rl, impg_amd.make_params(**kw))
            for i, (t, s, e) in enumerate(rl):
                assert got[i].tolist() == c.query(t, s, e, **kw).tolist(), (rank, i)
        dist.barrier()
        if rank == 0:
            print("fail ok world=%d" % world)
        del g
        comm.close()
        dist.destroy_process_group()
        return
    n_q = 0 if (rank == 1 and world == 3) else 18 + 5 * rank  # ranks own different numbers of queries; one owns none
    rl = random_ranges(100 + rank, n_q, n_seq, seq_len, max_len=3000, min_len=120)
    mask = {0: (seq_len, [(100, 2000), (5000, 9000)]), 2: (seq_len, [(0, 700)])}
    keep = np.array([1, 0, 1, 1, 0, 1, 1][:n_seq] + [1] * max(0, n_seq - 7), dtype=np.uint8)
    cases = [dict(), dict(transitive=True, max_depth=2), dict(transitive=True, max_depth=3, min_transitive_len=20),
             dict(transitive=True, max_depth=0, min_transitive_len=200, min_output_length=150),
             dict(transitive=True, dfs=True, max_depth=3, min_transitive_len=50),
             dict(transitive=True, max_depth=2, multi_impg=True), dict(transitive=True, dfs=True, max_depth=2, multi_impg=True),
             dict(min_identity=0.8), dict(transitive=True, max_depth=2, masked=True), dict(transitive=True, max_depth=2, subset=True)]
    for kw in cases:
        kw = dict(kw)
        m = mask if kw.pop("masked", False) else None
        sk = keep if kw.pop("subset", False) else None
        p = impg_amd.make_params(**kw)
        got = g.query_batch(rl, p, masked_regions=m, subset_keep=sk)
        total = 0
        for i, (t, s, e) in enumerate(rl):
            want = c.query(t, s, e, masked_regions=m, subset_keep=sk, **kw)
            assert got[i].tolist() == want.tolist(), (rank, i, kw)
            total += c.last_projection_count()
        if m is None and sk is None:
            st, cnt, ck = g.query_batch_stats(rl, p)
            tt = torch.tensor([st.projected, total], dtype=torch.int64)
            dist.all_reduce(tt)  # projections are counted where they are computed: compare the global sums
            assert int(tt[0]) == int(tt[1]), (rank, kw, tt.tolist())
            n_self = 1  # (no mask: one self interval per range)
            for i in range(len(rl)):
                assert int(cnt[i]) == len(got[i]) - n_self, (rank, i, kw)
    for kw in (dict(), dict(transitive=True, max_depth=2, min_transitive_len=40)):  # store_cigar: the slices' ops come home too
        res = g.query_batch(rl, impg_amd.make_params(store_cigar=True, **kw))
        for i, (t, s, e) in enumerate(rl):
            want, wcg = c.query_cigar(t, s, e, **kw)
            assert res[i].tolist() == want.tolist(), (rank, i, kw)
            got = res.cigars(i)
            assert [x.tolist() for x in got] == [x.tolist() for x in wcg], (rank, i, kw)
    # the row stream on a rank's shard: every chunk is a collective and the ranks bring different n (one none at all), so
    # they agree on the number of calls first; a rank whose consumer stops the stream keeps serving its peers' hops
    kw = dict(transitive=True, max_depth=2, min_transitive_len=40)
    p = impg_amd.make_params(**kw)
    seen = []
    proj = g.query_batch_stream(rl, lambda first, part: seen.append((first, [part[i].tolist() for i in range(len(part))])) and False,
                                p, chunk_ranges=5)
    flat = [rows for _, parts in seen for rows in parts]
    assert [f for f, _ in seen] == list(range(0, len(rl), 5)), (rank, [f for f, _ in seen])
    assert flat == [c.query(t, s, e, **kw).tolist() for t, s, e in rl], rank
    stopper = world - 1  # (the rank with the most ranges: its peers have long run out when it stops)
    calls = []
    g.query_batch_stream(rl, lambda first, part: calls.append(first) or (rank == stopper and len(calls) == 2), p, chunk_ranges=5)
    assert calls == (list(range(0, len(rl), 5))[:2] if rank == stopper else list(range(0, len(rl), 5))), (rank, calls)
    got = g.query_batch(rl, p)  # (and the handle still answers, in step with its peers)
    for i, (t, s, e) in enumerate(rl):
        assert got[i].tolist() == c.query(t, s, e, **kw).tolist(), (rank, i, "after the stopped stream")
    # the rank's shard saved and loaded back on the same communicator (one file per rank): same answers, no PAF needed
    saved = "%s.rank%dof%d.idx" % (paf_path, rank, world)
    g.save(saved)
    del g
    try:
        impg_amd.GpuImpg.load(saved)
        raise AssertionError("a shard must not open as a plain index")
    except impg_amd.ImpgGpuError as e:
        assert e.code == impg_amd.IMPG_E_INVALID
    g = impg_amd.GpuImpg.load(saved, device=device, comm=comm)
    assert g.shard_info()[:3] == (rank, world, lanes)
    assert (owner is None and g.shard_info()[3] is None) or g.shard_info()[3].tolist() == owner.tolist()
    g.set_option("chunk_ranges", 7)
    for kw in (dict(transitive=True, max_depth=3, min_transitive_len=20), dict(transitive=True, dfs=True, max_depth=2, multi_impg=True)):
        got = g.query_batch(rl, impg_amd.make_params(**kw))
        for i, (t, s, e) in enumerate(rl):
            assert got[i].tolist() == c.query(t, s, e, **kw).tolist(), (rank, i, kw, "loaded")
    os.unlink(saved)
    dist.barrier()
    if rank == 0:
        print("multi ok world=%d lanes=%d transport=%s" % (world, lanes, transport))
    del g
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
