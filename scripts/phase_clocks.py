#!/usr/bin/env python3
"""Where a wave of project_kernel spends its life (library built with -DIMPG_PHASE_CLOCKS, IMPG_GPU_LIB pointing at it):
s_memtime at the kernel's dependency boundaries, summed over the waves of every 64th block of one headline step.
usage: IMPG_GPU_LIB=impg_amd/libimpg_phase.so python scripts/phase_clocks.py [ranges]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tempfile  # noqa: E402

import numpy as np  # noqa: E402

import impg_amd  # noqa: E402

NAMES = ["lists + frontier record", "regroup (LDS sort)", "entry", "line headers", "end 1 (window + test)", "end 2 (window + test)",
         "stores issued + acknowledged", "final barrier + count"]


def main():
    n_ranges = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    lib = impg_amd.lib()
    fn = lib.impg_gpu_debug_phase_clocks
    fn.argtypes = [C.c_void_p]
    n_seq, seq_len, records = 200, 5_000_000, 1_000_000
    paf = os.path.join(tempfile.gettempdir(), "impg_synth_%d_seed42.paf" % records)
    if not os.path.exists(paf):
        impg_amd.synth_paf_text(paf, 42, records, n_seq=n_seq, seq_len=seq_len)
    g = impg_amd.GpuImpg.from_paf(paf)
    g.set_option("chunk_ranges", max(50000, n_ranges))
    g.set_option("pair_budget", 3 << 30)
    bed = impg_amd.synth_bed(7, n_ranges, n_seq=n_seq, seq_len=seq_len, range_len=5000)
    ranges = np.zeros(n_ranges, dtype=impg_amd.RANGE_DTYPE)
    ids = np.array([g.seq_id(impg_amd.synth_seq_name(t)) for t in range(n_seq)], dtype=np.uint32)
    ranges["target_id"], ranges["start"], ranges["end"] = ids[bed["target_id"]], bed["start"], bed["end"]
    params = impg_amd.make_params(transitive=True, max_depth=3)
    buf = (C.c_uint64 * 16)()
    g.query_batch_stats(ranges, params, counts=False, checksums=False)          # warm-up
    fn(C.cast(buf, C.c_void_p))                                                 # clear
    st, _, _ = g.query_batch_stats(ranges, params, counts=False, checksums=False)
    fn(C.cast(buf, C.c_void_p))
    n = buf[15]
    tot = sum(buf[i] for i in range(8))
    out = {"waves_sampled": int(n), "cycles_per_wave": tot / max(n, 1),
           "phases": [{"phase": NAMES[i], "cycles_per_wave": buf[i] / max(n, 1), "share": buf[i] / max(tot, 1)} for i in range(8)],
           "projected": int(st.projected), "ms_project": st.ms_project}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
