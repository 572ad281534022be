"""Config-5 shape (contiguous windows, -m 5) on the headline index: the counting form with and without per-range counts /
checksums, fused final level on / off -- stage times per call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import impg_amd
rec, ops, sl = impg_amd.synth_paf(42, 1_000_000)
g = impg_amd.GpuImpg.from_records(rec, ops, sl)
n = int(float(os.environ.get("WINDOWS", "4000")))
k = np.arange(n)
ranges = np.zeros(n, dtype=impg_amd.RANGE_DTYPE)
ranges["target_id"], ranges["start"] = k // 1000, (k % 1000) * 5000
ranges["end"] = ranges["start"] + 5000
g.set_option("pair_budget", 1 << 30)
g.set_option("chunk_ranges", int(os.environ.get("CHUNK", "2000")))
p5 = impg_amd.make_params(transitive=True, max_depth=5)
for fuse in (1, 0):
    g.set_option("fuse_final_level", fuse)
    for counts, cks in ((False, False), (True, False), (True, True), (False, False)):
        t0 = time.perf_counter()
        st, cnt, ck = g.query_batch_stats(ranges, p5, counts=counts, checksums=cks)
        print("fuse %d counts %d cksums %d: %.2f s wall, engine %.2f s (lookup %.0f project %.0f update %.0f ms), %.3g projections, %d project launches"
              % (fuse, counts, cks, time.perf_counter() - t0, st.ms_total / 1e3, st.ms_lookup, st.ms_project, st.ms_update, st.projected, st.project_launches), flush=True)
