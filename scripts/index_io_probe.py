#!/usr/bin/env python3
"""Start-up cost: building the headline index from its PAF vs loading the saved index (GPU box)."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import impg_amd
paf = os.path.join(tempfile.gettempdir(), "impg_synth_1000000_seed42.paf")
if not os.path.exists(paf):
    impg_amd.synth_paf_text(paf, 42, 1_000_000)
t = time.perf_counter(); g = impg_amd.GpuImpg.from_paf(paf); t_build = time.perf_counter() - t
out = os.path.join(tempfile.gettempdir(), "headline.impghbm")
t = time.perf_counter(); g.save(out); t_save = time.perf_counter() - t
t = time.perf_counter(); h = impg_amd.GpuImpg.load(out); t_load = time.perf_counter() - t
t = time.perf_counter(); h2 = impg_amd.GpuImpg.load(out); t_load2 = time.perf_counter() - t
print("PAF %.0f MB -> build %.2f s; saved index %.0f MB: save %.2f s, load %.2f s (again, page cache warm: %.2f s); %.2f GB in HBM"
      % (os.path.getsize(paf) / 1e6, t_build, os.path.getsize(out) / 1e6, t_save, t_load, t_load2, h.device_bytes() / 1e9))
r = impg_amd.synth_bed(7, 1000)
import numpy as np
ids = np.array([g.seq_id(impg_amd.synth_seq_name(i)) for i in range(200)], dtype=np.uint32)
rr = [(int(ids[a]), int(b), int(c)) for a, b, c in zip(r["target_id"], r["start"], r["end"])]
p = impg_amd.make_params(transitive=True, max_depth=2)
a, b = g.query_batch(rr, p), h.query_batch(rr, p)
assert all(a[i].tolist() == b[i].tolist() for i in range(len(rr)))
print("1000 transitive queries agree between the built and the loaded index")
os.remove(out)
