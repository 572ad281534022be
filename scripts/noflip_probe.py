#!/usr/bin/env python3
"""What would a projection kernel without the reversed-entry logic (I<->D swap, back-to-front tile walks) gain?
A UNIDIRECTIONAL index has no reversed entries, so a build whose kernel hard-wires swp = flip = false is still
exact there: run this once with the product library and once with IMPG_GPU_LIB=<that build> and compare
`project`.  (a second, pre-swapped copy of the ops for reversed entries.)"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import impg_amd
paf = os.path.join(tempfile.gettempdir(), "impg_synth_1000000_seed42.paf")
if not os.path.exists(paf):
    impg_amd.synth_paf_text(paf, 42, 1_000_000)
g = impg_amd.GpuImpg.from_paf(paf, bidirectional=False)
g.set_option("pair_budget", 1 << 30)
g.set_option("chunk_ranges", 50000)
ids = np.array([g.seq_id(impg_amd.synth_seq_name(i)) for i in range(200)], dtype=np.uint32)
bed = impg_amd.synth_bed(7, 100000)
r = np.zeros(len(bed), dtype=impg_amd.RANGE_DTYPE)
r["target_id"] = ids[bed["target_id"]]; r["start"], r["end"] = bed["start"], bed["end"]
p = impg_amd.make_params(transitive=True, max_depth=4)
g.query_batch_stats(r, p, counts=True, checksums=True)
st, cnt, ck = g.query_batch_stats(r, p, counts=True, checksums=True)
print("lib=%s unidirectional -x -m 4: projected %d pairs %d  lookup %.1f project %.1f update %.1f ms  checksum %x" %
      (os.environ.get("IMPG_GPU_LIB", "product"), st.projected, st.pairs, st.ms_lookup, st.ms_project, st.ms_update,
       int(np.bitwise_xor.reduce(ck))))
