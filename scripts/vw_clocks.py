#!/usr/bin/env python3
"""Where a wave of visited_update_wave_kernel spends its cycles on a config-5 batch (library built with -DIMPG_VW_CLOCKS,
IMPG_GPU_LIB pointing at it).  usage: IMPG_GPU_LIB=impg_amd/libimpg_vwclk.so python scripts/vw_clocks.py [windows]"""
import ctypes as C
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import impg_amd  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    lib = impg_amd.lib()
    fn = lib.impg_gpu_debug_vw_clocks
    fn.argtypes = [C.c_void_p]
    paf = os.path.join(tempfile.gettempdir(), "impg_synth_1000000_seed42.paf")
    if not os.path.exists(paf):
        impg_amd.synth_paf_text(paf, 42, 1_000_000)
    g = impg_amd.GpuImpg.from_paf(paf)
    g.set_option("chunk_ranges", 500)
    g.set_option("pair_budget", 1 << 30)
    ids = np.array([g.seq_id(impg_amd.synth_seq_name(t)) for t in range(200)], dtype=np.uint32)
    k = np.arange(n)
    ranges = np.zeros(n, dtype=impg_amd.RANGE_DTYPE)
    ranges["target_id"], ranges["start"] = ids[k // 1000], (k % 1000) * 5000
    ranges["end"] = ranges["start"] + 5000
    p = impg_amd.make_params(transitive=True, max_depth=5)
    buf = (C.c_uint64 * 16)()
    g.query_batch_stats(ranges[:500], p, counts=False, checksums=False)
    fn(C.cast(buf, C.c_void_p))
    st, _, _ = g.query_batch_stats(ranges, p, counts=False, checksums=False)
    fn(C.cast(buf, C.c_void_p))
    b = [int(x) for x in buf]
    groups = max(b[3], 1)
    out = {"groups": b[3], "cycles_per_group": b[15] / groups, "hits_per_group": b[4] / groups,
           "uncovered_share": b[5] / max(b[4], 1), "grown_or_isolated_share_of_uncovered": b[6] / max(b[5], 1),
           "sequential_share_of_uncovered": b[8] / max(b[5], 1), "groups_with_more_pieces_than_the_buffer": b[7] / groups,
           "cycles_share": {"batch head (load, coverage test, classes' tests)": b[0] / max(b[15], 1), "conflict loop": b[1] / max(b[15], 1),
                            "grown + isolated classes applied": b[2] / max(b[15], 1),
                            "sequential: lower bound + proximity": b[9] / max(b[15], 1), "sequential: walk (pieces)": b[10] / max(b[15], 1),
                            "sequential: plain insert (shift up)": b[11] / max(b[15], 1), "sequential: grow + merge forward (shift down)": b[12] / max(b[15], 1),
                            "pieces sorted": b[13] / max(b[15], 1), "pieces swept into ranges": b[14] / max(b[15], 1)},
           "cycles_per_sequential_hit": (b[9] + b[10] + b[11] + b[12]) / max(b[8], 1), "ms_update": st.ms_update}
    out["cycles_share"]["the rest (list in / out of LDS, group fetch)"] = 1.0 - sum(out["cycles_share"].values())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
