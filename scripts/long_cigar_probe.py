#!/usr/bin/env python3
"""Side measurement: long alignments (1 Mb, 20 000-op CIGARs -> 770 tiles, external checkpoints) instead of
the headline's 10 kb / 200 ops.  Same engine, -x -m 3, 20 000 query ranges of 5 kb."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import impg_amd

n_seq, seq_len = 40, 50_000_000
rec, ops, sl = impg_amd.synth_paf(42, 20_000, n_seq=n_seq, seq_len=seq_len, target_span=1_000_000, n_blocks=10_000)
t = time.time()
g = impg_amd.GpuImpg.from_records(rec, ops, sl)
print("index: %d records, %.1f M ops, built in %.1f s, %.2f GB" % (len(rec), ops.size / 1e6, time.time() - t, g.device_bytes() / 1e9))
g.set_option("pair_budget", 1 << 30)
bed = impg_amd.synth_bed(7, 20_000, n_seq=n_seq, seq_len=seq_len, range_len=5000)
r = np.zeros(len(bed), dtype=impg_amd.RANGE_DTYPE)
r["target_id"], r["start"], r["end"] = bed["target_id"], bed["start"], bed["end"]
for kw in (dict(), dict(transitive=True, max_depth=3)):
    p = impg_amd.make_params(**kw)
    g.query_batch_stats(r, p, counts=False, checksums=False)
    st, _, _ = g.query_batch_stats(r, p, counts=False, checksums=False)
    print("%s: pairs %d projected %d  lookup %.2f project %.2f update %.2f total %.2f ms -> %.2e projected/s (project kernel %.2e pairs/s)" %
          (kw or "plain", st.pairs, st.projected, st.ms_lookup, st.ms_project, st.ms_update, st.ms_total,
           st.projected / (st.ms_total * 1e-3), st.pairs / max(st.ms_project, 1e-9) * 1e3))
