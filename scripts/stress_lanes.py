#!/usr/bin/env python3
"""Timing-window stress for the multi-rank path (run on the GPU box): multi handles with several lanes and tiny
chunks, batches whose walks end at their first hop -- so that lanes finish and hand back their engines while others
are still starting, the window in which per-lease state was once lost (seed 72686) -- through every entry
point, thousands of times, against answers the oracle gave once.  usage: stress_lanes.py [seconds]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import impg_amd
from oracle import oracle as o
from tests.paf_gen import random_paf, random_ranges
from tests.test_gpu_fullsize import checksum

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
tmp = tempfile.mkdtemp()
text, _ = random_paf(4242, 400, n_seq=7, seq_len=20000, self_aln=True)
path = os.path.join(tmp, "s.paf")
open(path, "w").write(text)
c = o.OracleIndex(paf_paths=[path], preparse=True)
seq_len = int(c.seq_len(0))
full_mask = {s: (seq_len, [(0, seq_len)]) for s in range(c.num_seqs())}
part_mask = {0: (seq_len, [(100, 2000), (5000, 9000)]), 3: (seq_len, [(0, 700)])}
keep = np.array([0, 1, 0, 1, 1, 0, 1], dtype=np.uint8)
short = random_ranges(1, 11, c.num_seqs(), 20000, max_len=400, min_len=120)     # below min_transitive_len: nothing to expand
longer = random_ranges(2, 7, c.num_seqs(), 20000, max_len=2500, min_len=700)
SHAPES = [  # (ranges, params, mask, subset)
    (short, dict(), None, None),
    (short, dict(transitive=True, max_depth=1, min_transitive_len=500), None, None),
    (short, dict(transitive=True, dfs=True, max_depth=1, min_transitive_len=500), None, None),
    (short, dict(transitive=True, max_depth=1, min_transitive_len=500), full_mask, None),
    (short, dict(transitive=True, dfs=True, max_depth=1, min_transitive_len=500), part_mask, None),
    (short, dict(transitive=True, max_depth=1, min_transitive_len=500, multi_impg=True), None, None),
    (short, dict(transitive=True, max_depth=1, min_transitive_len=500), None, keep),
    (longer, dict(transitive=True, max_depth=2, min_transitive_len=100), part_mask, keep),
    (longer, dict(transitive=True, dfs=True, max_depth=2, min_transitive_len=100, multi_impg=True), None, None),
]
want = []
for rl, kw, m, k in SHAPES:
    rows = [c.query(t, s, e, masked_regions=m, subset_keep=k, **kw) for (t, s, e) in rl]
    proj = 0
    for (t, s, e) in rl:
        c.query(t, s, e, masked_regions=m, subset_keep=k, **kw)
        proj += c.last_projection_count()
    want.append(([r.tolist() for r in rows], proj, [len(r) - 1 for r in rows], [checksum(r[1:]) for r in rows]))
t_end = time.time() + budget
n_batches = 0
rng = np.random.default_rng(7)
while time.time() < t_end:
    world, lanes, chunk = int(rng.choice([2, 3, 4])), int(rng.choice([2, 3])), int(rng.choice([1, 2, 3]))
    g = impg_amd.GpuImpg.from_paf(path, devices=[0] * world, lanes=lanes)
    g.set_option("chunk_ranges", chunk)
    for rep in range(150):
        for (rl, kw, m, k), (rows, proj, cnt, ck) in zip(SHAPES, want):
            p = impg_amd.make_params(**kw)
            got = g.query_batch(rl, p, masked_regions=m, subset_keep=k)
            assert got.projected == proj, ("projected", world, lanes, chunk, rep, kw)
            for i in range(len(rl)):
                assert got[i].tolist() == rows[i], ("rows", world, lanes, chunk, rep, kw, i, rl[i], got[i].tolist(), rows[i])
            if m is None and k is None:  # the counting entry point takes neither
                st, gc, gk = g.query_batch_stats(rl, p)
                assert st.projected == proj and gc.tolist() == cnt and [int(x) for x in gk] == ck, ("stats", world, lanes, chunk, rep, kw)
            n_batches += 1
    del g
print("stress ok: %d batches" % n_batches)
