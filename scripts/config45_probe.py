#!/usr/bin/env python3
"""BASELINE configs 4 and 5 on one MI355X: sizes, build time, query time.
  config45_probe.py 4 <records> [n_seq] [ranges]     HPRC-scale index from impg_synth_paf records (no PAF text)
  config45_probe.py 5 <windows> [records] [n_seq]    contiguous 5 kb windows, -x -m 5"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import impg_amd  # noqa: E402

which = sys.argv[1]
t0 = time.time()


def log(m):
    print("[%7.1fs] %s" % (time.time() - t0, m), file=sys.stderr, flush=True)


if which == "4":
    records = int(float(sys.argv[2]))
    n_seq = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
    n_ranges = int(sys.argv[4]) if len(sys.argv) > 4 else 100000
    seq_len = 5_000_000
    rec, ops, sl = impg_amd.synth_paf(42, records, n_seq=n_seq, seq_len=seq_len)
    log("synth: %d records, %.1f GB of ops" % (records, ops.nbytes / 1e9))
    tb = time.time()
    g = impg_amd.GpuImpg.from_records(rec, ops, sl)
    build_s = time.time() - tb
    log("index: %.1f s, %.1f GB in HBM, %d entries" % (build_s, g.device_bytes() / 1e9, g.num_entries()))
    del rec, ops
    ranges = impg_amd.synth_bed(7, n_ranges, n_seq=n_seq, seq_len=seq_len, range_len=5000)
    p = impg_amd.make_params(transitive=True, max_depth=3)
    g.set_option("chunk_ranges", 25000)
    g.set_option("pair_budget", 1 << 30)
    st, _, _ = g.query_batch_stats(ranges, p, counts=False, checksums=False)
    tq = time.time()
    st, cnt, ck = g.query_batch_stats(ranges, p)
    dt = time.time() - tq
    print(json.dumps({"config": 4, "records": records, "n_seq": n_seq, "ranges": n_ranges, "index_GB": g.device_bytes() / 1e9,
                      "build_s": build_s, "query_s": dt, "projected": st.projected, "projected_per_s": st.projected / dt,
                      "levels": st.levels, "ms": [st.ms_lookup, st.ms_project, st.ms_update]}))
else:
    windows = int(float(sys.argv[2]))
    records = int(float(sys.argv[3])) if len(sys.argv) > 3 else 5_000_000
    n_seq = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
    seq_len = 5_000_000
    rec, ops, sl = impg_amd.synth_paf(42, records, n_seq=n_seq, seq_len=seq_len)
    tb = time.time()
    g = impg_amd.GpuImpg.from_records(rec, ops, sl)
    build_s = time.time() - tb
    log("index: %.1f s, %.1f GB" % (build_s, g.device_bytes() / 1e9))
    del rec, ops
    per_seq = seq_len // 5000
    ranges = np.zeros(windows, dtype=impg_amd.RANGE_DTYPE)
    k = np.arange(windows)
    ranges["target_id"] = k // per_seq
    ranges["start"] = (k % per_seq) * 5000
    ranges["end"] = ranges["start"] + 5000
    for depth in [int(x) for x in os.environ.get("DEPTHS", "3,5").split(",")]:
        p = impg_amd.make_params(transitive=True, max_depth=depth)
        g.set_option("chunk_ranges", int(os.environ.get("CHUNK", "2000" if depth == 5 else "25000")))
        g.set_option("pair_budget", 1 << 30)
        tq = time.time()
        st, cnt, ck = g.query_batch_stats(ranges, p)
        dt = time.time() - tq
        log("depth %d: %.2f s, %d projected (%.3g /s), per window %.0f" % (depth, dt, st.projected, st.projected / dt, st.projected / windows))
        print(json.dumps({"config": 5, "windows": windows, "records": records, "n_seq": n_seq, "depth": depth, "query_s": dt,
                          "projected": st.projected, "projected_per_s": st.projected / dt, "levels": st.levels,
                          "ms": [st.ms_lookup, st.ms_project, st.ms_update]}))
