#!/usr/bin/env python3
"""Where a wave of visited_update_kernel spends its life (library built with -DIMPG_VU_CLOCKS, IMPG_GPU_LIB pointing at
it): the cycle counter at the kernel's phase boundaries, summed over the waves of one headline step.
usage: IMPG_GPU_LIB=impg_amd/libimpg_vuclk.so python scripts/vu_clocks.py [ranges]"""
import ctypes as C
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import impg_amd  # noqa: E402

NAMES = ["round 1: the group's record", "round 2: first hits, first ranges, clamp length", "in-place path (cap > LDS column)", "list into LDS",
         "replay", "list written out", "pieces sorted, merged, stored"]


def main():
    n_ranges = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    lib = impg_amd.lib()
    fn = lib.impg_gpu_debug_vu_clocks
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
    params = impg_amd.make_params(transitive=True, max_depth=int(os.environ.get("DEPTH", "3")))
    buf = (C.c_uint64 * 16)()
    g.query_batch_stats(ranges, params, counts=False, checksums=False)          # warm-up
    fn(C.cast(buf, C.c_void_p))                                                 # clear
    st, _, _ = g.query_batch_stats(ranges, params, counts=False, checksums=False)
    fn(C.cast(buf, C.c_void_p))
    n = max(buf[8], 1)
    tot = sum(buf[i] for i in range(7))
    out = {"waves": int(buf[8]), "cycles_per_wave": tot / n,
           "phases": [{"phase": NAMES[i], "cycles_per_wave": buf[i] / n, "share": buf[i] / max(tot, 1)} for i in range(7)],
           "waves_with_an_in_place_group": int(buf[9]), "mean_longest_replay_of_a_wave": buf[10] / n, "mean_hits_per_wave": buf[11] / n,
           "mean_hits_of_the_in_place_group": buf[12] / max(buf[9], 1), "waves_with_pieces_beyond_registers": int(buf[13]),
           "in_place_cycles_per_such_wave": buf[14] / max(buf[9], 1), "ms_update": st.ms_update}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
