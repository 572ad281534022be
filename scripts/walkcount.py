import os, sys, tempfile
sys.path.insert(0, "/root/repo")
import numpy as np
import impg_amd
paf = os.path.join(tempfile.gettempdir(), "impg_synth_1000000_seed42.paf")
if not os.path.exists(paf):
    impg_amd.synth_paf_text(paf, 42, 1_000_000)
g = impg_amd.GpuImpg.from_paf(paf)
bed = impg_amd.synth_bed(7, 20000)
ids = np.array([g.seq_id(impg_amd.synth_seq_name(i)) for i in range(200)], dtype=np.uint32)
r = np.zeros(len(bed), dtype=impg_amd.RANGE_DTYPE)
r["target_id"] = ids[bed["target_id"]]; r["start"], r["end"] = bed["start"], bed["end"]
for kw in [dict(transitive=True, max_depth=3), dict()]:
    st, _, _ = g.query_batch_stats(r, impg_amd.make_params(**kw), counts=False, checksums=False)
    print(kw, "pairs", st.pairs, "projected", st.projected & ((1 << 40) - 1), "walked", st.projected >> 40, "project ms", st.ms_project)
