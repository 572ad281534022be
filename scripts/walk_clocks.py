"""IMPG_WALK_DEBUG=1 python scripts/walk_clocks.py: the walk kernel's phase clocks for a few single calls on the headline
index (BFS -m 3 in the grid form, masked, DFS -m 3)."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("IMPG_WALK_DEBUG", "1")
import impg_amd
paf = os.path.join(tempfile.gettempdir(), "impg_synth_1000000_seed42.paf")
if not os.path.exists(paf):
    impg_amd.synth_paf_text(paf, 42, 1_000_000)
g = impg_amd.GpuImpg.from_paf(paf)
bed = impg_amd.synth_bed(7, 64)
ranges = [(g.seq_id(impg_amd.synth_seq_name(int(t))), int(s), int(e)) for t, s, e in zip(bed["target_id"], bed["start"], bed["end"])]
rng = np.random.default_rng(5)
mask = {}
for sid in range(g.num_seqs()):
    cuts = np.sort(rng.choice(1000, size=200, replace=False)) * 5000
    merged = []
    for a in cuts:
        if merged and merged[-1][1] >= a: merged[-1] = (merged[-1][0], int(a) + 5000)
        else: merged.append((int(a), int(a) + 5000))
    mask[sid] = (5_000_000, merged)
mask = impg_amd.prepare_mask(mask)
for depth in (2, 3):
    print("--- masked bfs -m", depth, file=sys.stderr, flush=True)
    g.query_batch(ranges[1:2], impg_amd.make_params(transitive=True, max_depth=depth), copy=False, masked_regions=mask)
    g.query_batch(ranges[1:2], impg_amd.make_params(transitive=True, max_depth=depth), copy=False, masked_regions=mask)
for members in (0, 1):
    g.set_option("walk_kernel", 2)
    g.set_option("walk_members", members)
    for k in range(3):
        print("--- bfs -m 3 members", members, "call", k, file=sys.stderr, flush=True)
        g.query_batch(ranges[k:k + 1], impg_amd.make_params(transitive=True, max_depth=3), copy=False)
print("--- dfs -m 3", file=sys.stderr, flush=True)
g.query_batch(ranges[0:1], impg_amd.make_params(transitive=True, dfs=True, max_depth=3), copy=False)
