"""Per-call latency of query_transitive_bfs -m 3 (and DFS -m 3) on the headline index: the batch engine (walk_kernel 0),
the one-workgroup walk (walk_kernel 2, walk_members 1) and the walk's grid form (walk_members 2 .. 64), unmasked and
under a partition-style mask; counts must agree."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import impg_amd
paf = os.path.join(tempfile.gettempdir(), "impg_synth_1000000_seed42.paf")
if not os.path.exists(paf):
    impg_amd.synth_paf_text(paf, 42, 1_000_000)
g = impg_amd.GpuImpg.from_paf(paf)
bed = impg_amd.synth_bed(7, 4096)
ranges = [(g.seq_id(impg_amd.synth_seq_name(int(t))), int(s), int(e)) for t, s, e in zip(bed["target_id"], bed["start"], bed["end"])]
g.set_option("prewarm_walk", 1)
# a partition-style mask: on every sequence a few hundred earlier windows are taken
rng = np.random.default_rng(5)
mask = {}
for sid in range(g.num_seqs()):
    L = 5_000_000
    cuts = np.sort(rng.choice(L // 5000, size=200, replace=False)) * 5000
    mask[sid] = (L, [(int(c), int(c) + 5000) for c in cuts if not any(c == r[1] for r in [])])
    # (touching windows are merged, as a SortedRanges would hold them)
    merged = []
    for a, b in mask[sid][1]:
        if merged and merged[-1][1] >= a: merged[-1] = (merged[-1][0], b)
        else: merged.append((a, b))
    mask[sid] = (L, merged)
mask = impg_amd.prepare_mask(mask)  # (the dict -> C layout conversion is ~3 ms of Python per call otherwise)
reps = int(os.environ.get("REPS", "200"))
def timed(label, fn):
    for k in range(5): fn(k)
    t0 = time.perf_counter()
    rows = 0
    for k in range(reps): rows += fn(k + 5)
    dt = (time.perf_counter() - t0) / reps
    print("%-58s %8.1f us per call, %7.0f rows per call" % (label, dt * 1e6, rows / reps), flush=True)
    return rows
for depth in (3, 2):
    want = None
    for wk, members in ((0, 0), (2, 1), (1, 4), (1, 0)):
        g.set_option("walk_kernel", wk); g.set_option("walk_members", members)
        p = impg_amd.make_params(transitive=True, max_depth=depth)
        rows = timed("bfs -m %d walk_kernel %d members %2d" % (depth, wk, members), lambda k: g.query_batch(ranges[k:k + 1], p, copy=False).total)
        assert want is None or rows == want, (rows, want)
        want = rows
        rows_m = timed("bfs -m %d walk_kernel %d members %2d masked" % (depth, wk, members),
                       lambda k: g.query_batch(ranges[k:k + 1], p, copy=False, masked_regions=mask).total)
    print("walk launches", g.counter("walk_launches"), "fallbacks", g.counter("walk_fallbacks"))
g.set_option("walk_kernel", 1); g.set_option("walk_members", 0)
p = impg_amd.make_params(transitive=True, max_depth=3)
for nb in (8, 64):
    timed("bfs -m 3 grid batch %d" % nb, lambda k: g.query_batch(ranges[(k * nb) % 4000:(k * nb) % 4000 + nb], p, copy=False).total)
g.set_option("walk_kernel", 0)
for nb in (8, 64):
    timed("bfs -m 3 batch engine batch %d" % nb, lambda k: g.query_batch(ranges[(k * nb) % 4000:(k * nb) % 4000 + nb], p, copy=False).total)
reps = 10
pd = impg_amd.make_params(transitive=True, dfs=True, max_depth=3)
g.set_option("walk_kernel", 1)
timed("dfs -m 3 walk", lambda k: g.query_batch(ranges[k:k + 1], pd, copy=False).total)
timed("dfs -m 3 walk masked", lambda k: g.query_batch(ranges[k:k + 1], pd, copy=False, masked_regions=mask).total)
print("walk launches", g.counter("walk_launches"), "fallbacks", g.counter("walk_fallbacks"))
