import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import impg_amd
paf = os.path.join(tempfile.gettempdir(), "impg_synth_1000000_seed42.paf")
if not os.path.exists(paf):
    impg_amd.synth_paf_text(paf, 42, 1_000_000)
g = impg_amd.GpuImpg.from_paf(paf)
bed = impg_amd.synth_bed(7, 4096)
ranges = np.zeros(len(bed), dtype=impg_amd.RANGE_DTYPE)
ranges["target_id"] = [g.seq_id(impg_amd.synth_seq_name(int(t))) for t in bed["target_id"]]
ranges["start"], ranges["end"] = bed["start"], bed["end"]
p = impg_amd.make_params(transitive=True, max_depth=3)
for wk in (2, 0):
    g.set_option("walk_kernel", wk)
    for k in range(3):
        t0 = time.perf_counter()
        st, cnt, ck = g.query_batch_stats(ranges[k:k+1], p)
        print("walk_kernel", wk, "bfs -m 3 n=1: %.1f us, %d projections" % ((time.perf_counter() - t0) * 1e6, st.projected), flush=True)
