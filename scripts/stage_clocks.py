#!/usr/bin/env python3
"""Where a block of project_staged_kernel spends its life (library built with -DIMPG_PHASE_CLOCKS, IMPG_GPU_LIB pointing
at it): cycle counter at the block's phase boundaries, summed over every 16th block of one headline step, and how many
blocks did not stage (span of entries wider than STG_ECAP).
usage: IMPG_GPU_LIB=impg_amd/libimpg_phase.so python scripts/stage_clocks.py [ranges]"""
import ctypes as C
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import impg_amd  # noqa: E402

NAMES = ["the block's ranges to LDS + span of entries", "stage entries + prefix lines", "projections (all turns)"]


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
    n, nf = buf[15], buf[14]
    out = {"staged_blocks_sampled": int(n), "unstaged_blocks_sampled": int(nf),
           "span_of_staged_block": buf[11] / max(n, 1), "span_of_unstaged_block": buf[12] / max(nf, 1),
           "cycles_per_unstaged_block": buf[13] / max(nf, 1),
           "pairs_per_staged_block": buf[9] / max(n, 1), "pairs_listed_for_the_general_form": int(buf[8]),
           "listed_share_of_pairs": buf[8] / max(buf[9], 1), "staged_blocks_wider_than_the_buffer": buf[10] / max(n, 1),
           "phases": [{"phase": NAMES[i], "cycles_per_block": buf[i] / max(n, 1)} for i in range(3)],
           "projected": int(st.projected), "ms_project": st.ms_project}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
