#!/usr/bin/env python3
"""How much of the headline workload's result depends on the visit order of overlapping entries?
The coitrees visit order is a restatement no reference test pins; this runs the headline
batch (1M-record PAF, 100k ranges, -x -m 3) under IMPG_ORDER_COITREES and IMPG_ORDER_SORTED and compares, per
range, the number of result rows and an order-independent checksum of them.  Whatever a wrong recollection of
the coitrees order could change is bounded by what a completely different order changes."""
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import impg_amd  # noqa: E402

records = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_ranges = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
n_seq, seq_len = 200, 5_000_000
paf = os.path.join(tempfile.gettempdir(), "impg_synth_%d_seed42.paf" % records)
if not os.path.exists(paf):
    impg_amd.synth_paf_text(paf, 42, records, n_seq=n_seq, seq_len=seq_len)
p = impg_amd.make_params(transitive=True, max_depth=3)
out = {}
for name, order in (("coitrees", impg_amd.ORDER_COITREES), ("sorted", impg_amd.ORDER_SORTED)):
    g = impg_amd.GpuImpg.from_paf(paf, order=order)
    g.set_option("chunk_ranges", 50000)
    g.set_option("pair_budget", 1 << 30)
    bed = impg_amd.synth_bed(7, n_ranges, n_seq=n_seq, seq_len=seq_len, range_len=5000)
    ranges = np.zeros(n_ranges, dtype=impg_amd.RANGE_DTYPE)
    ranges["target_id"] = [g.seq_id(impg_amd.synth_seq_name(int(t))) for t in bed["target_id"]]
    ranges["start"], ranges["end"] = bed["start"], bed["end"]
    st, cnt, ck = g.query_batch_stats(ranges, p)
    out[name] = (st.projected, cnt, ck)
    del g
(pa, ca, ka), (pb, cb, kb) = out["coitrees"], out["sorted"]
res = {"workload": "%d-record synthetic PAF, %d ranges, -x -m 3" % (records, n_ranges),
       "projected_coitrees_order": int(pa), "projected_sorted_order": int(pb),
       "ranges_with_different_row_count": int((ca != cb).sum()),
       "ranges_with_different_row_set": int((ka != kb).sum()),
       "sum_abs_row_count_difference": int(np.abs(ca.astype(np.int64) - cb.astype(np.int64)).sum()),
       "rows_total": int(ca.sum())}
print(json.dumps(res))
