#!/usr/bin/env python3
"""Per-chunk timing of impg_gpu_query_batch_stream on the headline batch (engine / assemble seconds of every chunk, the
consumer's wall clock between chunks).  usage: python scripts/stream_probe.py [ranges] [chunk]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import impg_amd

n_ranges = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
n_seq, seq_len, records = 200, 5_000_000, 1_000_000
paf = os.path.join(tempfile.gettempdir(), "impg_synth_%d_seed42.paf" % records)
if not os.path.exists(paf):
    impg_amd.synth_paf_text(paf, 42, records, n_seq=n_seq, seq_len=seq_len)
g = impg_amd.GpuImpg.from_paf(paf)
bed = impg_amd.synth_bed(7, n_ranges, n_seq=n_seq, seq_len=seq_len, range_len=5000)
ranges = np.zeros(n_ranges, dtype=impg_amd.RANGE_DTYPE)
ids = np.array([g.seq_id(impg_amd.synth_seq_name(t)) for t in range(n_seq)], dtype=np.uint32)
ranges["target_id"], ranges["start"], ranges["end"] = ids[bed["target_id"]], bed["start"], bed["end"]
params = impg_amd.make_params(transitive=True, max_depth=3)
for rep in range(2):
    t0 = time.perf_counter()
    marks = []
    def consume(first, part):
        marks.append((time.perf_counter() - t0, first, part.total) + part.timing())
        return False
    g.query_batch_stream(ranges, consume, params, chunk_ranges=chunk)
    dt = time.perf_counter() - t0
    print("rep %d: %.3f s" % (rep, dt))
    for m in marks:
        print("  at %.3f s first=%d rows=%d engine=%.3f assemble=%.3f" % m)
