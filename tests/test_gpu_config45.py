"""BASELINE configs 4 and 5 on one MI355X.

Config 4 (HPRC scale): the synthetic generator at 20 000 sequences, index built straight from impg_synth_paf
records (no PAF text).  Default 2 x 10^7 records in the suite (the driver's run has a time limit; scripts/final_r5.sh
runs this test at IMPG_CONFIG4_RECORDS=5e7 -- 162 GB of index -- and records it under profiles/; 10^8 fits one 288 GB MI355X).  The oracle cannot hold an index of this size in the time a test has, so the checks at full size
are: (a) non-transitive results of a sample of ranges against a brute-force scan of the record arrays with every
projection done by the oracle's project_target_range_through_alignment; (b) -x -m 3 over 100 000 ranges is
invariant under re-chunking; (c) the same batch on the index sharded three ways (multi handle) gives identical
per-range counts and checksums.  Transitive exactness vs the oracle at this SHAPE is tests/test_gpu_fullsize.py
(1M records).

Config 5 (partition-style tiling): contiguous 5 kb windows end to end, -x -m 5.
"""
import os

import numpy as np
import pytest

import impg_amd
from oracle import oracle as o
from tests.test_gpu_fullsize import checksum

pytestmark = pytest.mark.gpu

N_SEQ4, SEQ_LEN = 20000, 5_000_000


def brute_force_query(rec, ops, t, s, e):
    """Impg::query (impg.rs:1852-1928) on raw records: every forward entry of target t and every reversed entry
    (records whose QUERY is t: axes swapped, CIGAR inverted, impg.rs:1584-1607, :144-156) whose closed interval
    meets [s, e], projected through the oracle."""
    rows = []
    fw = np.nonzero((rec["target_id"] == t) & (rec["target_start"] <= e) & (rec["target_end"] >= s))[0]
    for i in fw:
        r = rec[i]
        cg = ops[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["cigar_len"])]
        pr = o.project((s, e), (int(r["target_start"]), int(r["target_end"]), int(r["query_start"]), int(r["query_end"]),
                               int(r["strand"])), cg)
        if pr is not None:
            rows.append((int(r["query_id"]), pr[0], pr[1], t, pr[3], pr[4]))
    rv = np.nonzero((rec["query_id"] == t) & (rec["target_id"] != t) & (rec["query_start"] <= e) & (rec["query_end"] >= s))[0]
    for i in rv:
        r = rec[i]
        cg = o.invert_cigar(ops[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["cigar_len"])], int(r["strand"]))
        pr = o.project((s, e), (int(r["query_start"]), int(r["query_end"]), int(r["target_start"]), int(r["target_end"]),
                               int(r["strand"])), cg)
        if pr is not None:
            rows.append((int(r["target_id"]), pr[0], pr[1], t, pr[3], pr[4]))
    return np.array(rows, dtype=impg_amd.INTERVAL_DTYPE)


CODE = "=XIDM"


def oracle_on_subset(rec, ops, ranges_2hop, n_seq, seq_len):
    # An oracle index that answers a walk exactly like one over ALL records would: every record that overlaps any range the
    # walk can look up, behind one 1-bp dummy alignment per pair of sequences that pins the sequence ids to the generator's
    # (the oracle numbers sequences in first-seen order, and ids order the frontier).  ranges_2hop: (seq, start, end) supersets
    # of every range looked up.  Visit order: the oracle's sorted policy (a subset keeps a target's start order, not its
    # coitrees ranks), so the engine's index is built with IMPG_ORDER_SORTED too.
    pick = np.zeros(len(rec), dtype=bool)
    for (t, s, e) in ranges_2hop:
        pick |= (rec["target_id"] == t) & (rec["target_start"] <= e) & (rec["target_end"] >= s)
        pick |= (rec["query_id"] == t) & (rec["query_start"] <= e) & (rec["query_end"] >= s)
    name = impg_amd.synth_seq_name
    lines = []
    for k in range(0, n_seq, 2):  # ids 0, 1, 2, ... in first-seen order; the 1-bp records sit at the very end of the sequences
        lines.append("%s\t%d\t%d\t%d\t+\t%s\t%d\t%d\t%d\t1\t1\t60\tcg:Z:1=" %
                     (name(k), seq_len, seq_len - 1, seq_len, name(k + 1), seq_len, seq_len - 1, seq_len))
    for i in np.nonzero(pick)[0]:
        r = rec[i]
        cg = ops[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["cigar_len"])]
        text = "".join("%d%s" % (int(v) & ((1 << 29) - 1), CODE[int(v) >> 29]) for v in cg)
        lines.append("%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t1\t1\t60\tcg:Z:%s" %
                     (name(int(r["query_id"])), seq_len, int(r["query_start"]), int(r["query_end"]), "+-"[int(r["strand"])],
                      name(int(r["target_id"])), seq_len, int(r["target_start"]), int(r["target_end"]), text))
    return o.OracleIndex(paf_text="\n".join(lines) + "\n", preparse=True), int(pick.sum())


def test_config4_hprc_scale_index():
    records = int(float(os.environ.get("IMPG_CONFIG4_RECORDS", "2e7")))
    rec, ops, sl = impg_amd.synth_paf(42, records, n_seq=N_SEQ4, seq_len=SEQ_LEN)
    g = impg_amd.GpuImpg.from_records(rec, ops, sl)
    assert g.num_entries() == 2 * records and g.num_seqs() == N_SEQ4
    n_ranges = 100_000
    ranges = impg_amd.synth_bed(7, n_ranges, n_seq=N_SEQ4, seq_len=SEQ_LEN, range_len=5000)
    # (a) non-transitive, a sample against the brute-force scan
    plain = impg_amd.make_params()
    st0, cnt0, ck0 = g.query_batch_stats(ranges, plain)
    rng = np.random.default_rng(4)
    sample = sorted(set(rng.integers(0, n_ranges, 10).tolist()) | {0, n_ranges - 1})
    full = g.query_batch(ranges[sample], plain)
    seen = 0
    for k, i in enumerate(sample):
        r = ranges[i]
        want = brute_force_query(rec, ops, int(r["target_id"]), int(r["start"]), int(r["end"]))
        got = full[k][1:]  # the self interval first
        assert sorted(got.tolist()) == sorted(want.tolist()), i
        assert int(cnt0[i]) == len(want) and int(ck0[i]) == checksum(want), i
        seen += len(want)
    assert seen > 30
    # (a') transitive, two hops, exact: the oracle on every record the walk can touch (found by brute force over the
    # arrays), against an index of ALL records under the sorted visit order
    gs = impg_amd.GpuImpg.from_records(rec, ops, sl, order=impg_amd.ORDER_SORTED)
    o.set_sorted_visits(True)
    try:
        kw = dict(transitive=True, max_depth=2)
        for i in sample[:3]:
            r = ranges[i]
            t, s0, e0 = int(r["target_id"]), int(r["start"]), int(r["end"])
            hop1 = gs.query_batch([(t, s0, e0)], plain)[0]
            look = [(t, s0, e0)] + [(int(h["query_id"]), min(int(h["q_first"]), int(h["q_last"])), max(int(h["q_first"]), int(h["q_last"])))
                                    for h in hop1[1:]]
            c2, n_sub = oracle_on_subset(rec, ops, look, N_SEQ4, SEQ_LEN)
            assert c2.num_seqs() == N_SEQ4 and c2.seq_name(t) == impg_amd.synth_seq_name(t)
            want = c2.query(t, s0, e0, **kw)
            got = gs.query_batch([(t, s0, e0)], impg_amd.make_params(**kw))[0]
            assert got.tolist() == want.tolist(), (i, n_sub)
            assert len(want) > len(hop1) and SEQ_LEN - 1 not in set(want["t_first"].tolist())  # (no dummy record was reached)
    finally:
        o.set_sorted_visits(False)
    del gs
    # (b) -x -m 3: chunking does not matter
    p = impg_amd.make_params(transitive=True, max_depth=3)
    g.set_option("pair_budget", 1 << 30)
    g.set_option("chunk_ranges", 25000)
    st1, cnt1, ck1 = g.query_batch_stats(ranges, p)
    g.set_option("chunk_ranges", 7777)
    st2, cnt2, ck2 = g.query_batch_stats(ranges, p)
    assert st1.levels == 3 and st1.projected == st2.projected == int(cnt1.sum()) > 100 * n_ranges
    assert (cnt1 == cnt2).all() and (ck1 == ck2).all()
    # (c) the same batch on three shards
    del g, full
    m = impg_amd.GpuImpg.from_records(rec, ops, sl, devices=[0, 0, 0], lanes=2)
    m.set_option("pair_budget", 1 << 30)
    m.set_option("chunk_ranges", 10000)
    st3, cnt3, ck3 = m.query_batch_stats(ranges, p)
    assert st3.projected == st1.projected and (cnt3 == cnt1).all() and (ck3 == ck1).all()
    own = m.shard_info()[3]
    load = np.bincount(own, weights=np.bincount(np.concatenate([rec["target_id"], rec["query_id"]]), minlength=N_SEQ4), minlength=3)
    assert load.max() / load.mean() < 1.01  # bin-packed by entry count


def test_config5_window_tiling_depth5():
    """Partition-style tiling on the headline index (1M records, 200 sequences x 5 Mb): contiguous 5 kb windows laid
    end to end over whole sequences, -x -m 5.  At this depth every window's closure saturates (each reaches most of
    the 2M entries), which is what the configuration is for: hit volume, long visited lists, frontiers of long merged
    ranges.  Checked: an oracle sample by count / checksum on a small tiling (the oracle needs seconds per window at
    depth 5 even there), and on the large tiling that re-chunking changes nothing and that a window's closure at
    depth 5 contains its closure at depth 3."""
    # small tiling, exact: 20 000-record index over 20 sequences x 1 Mb, the first 40 windows of two sequences
    import tempfile
    paf = os.path.join(tempfile.gettempdir(), "impg_c5_small.paf")
    impg_amd.synth_paf_text(paf, 42, 20000, n_seq=20, seq_len=1_000_000)
    gs = impg_amd.GpuImpg.from_paf(paf)
    cs = o.OracleIndex(paf_paths=[paf], preparse=True)
    small = [(gs.seq_id(impg_amd.synth_seq_name(t)), 5000 * k, 5000 * (k + 1)) for t in (0, 7) for k in range(20)]
    p5 = impg_amd.make_params(transitive=True, max_depth=5)
    st, cnt, ck = gs.query_batch_stats(small, p5)
    for i in (0, 13, 39):
        want = cs.query(*small[i], transitive=True, max_depth=5)[1:]
        assert int(cnt[i]) == len(want) and int(ck[i]) == checksum(want), i
    assert int(cnt.min()) > 10_000
    del gs, cs
    # large tiling on the headline index
    n_windows = int(float(os.environ.get("IMPG_CONFIG5_WINDOWS", "4e4")))  # (the suite's default; scripts/final_r5.sh runs 2e5 -- a fifth of BASELINE config 5's 10^6, ~75 s with the per-range counts and checksums this test reads -- and bench.py --workload config5 the 10^6)
    rec, ops, sl = impg_amd.synth_paf(42, 1_000_000)
    g = impg_amd.GpuImpg.from_records(rec, ops, sl)
    per_seq = SEQ_LEN // 5000
    k = np.arange(n_windows)
    ranges = np.zeros(n_windows, dtype=impg_amd.RANGE_DTYPE)
    ranges["target_id"], ranges["start"] = k // per_seq, (k % per_seq) * 5000
    ranges["end"] = ranges["start"] + 5000
    g.set_option("pair_budget", 1 << 30)
    g.set_option("chunk_ranges", 2000)
    import time
    t0 = time.perf_counter()
    st1, cnt1, ck1 = g.query_batch_stats(ranges, p5)
    print("config 5: %d windows -m 5 in %.1f s (engine %.1f s), %.3g projections" % (n_windows, time.perf_counter() - t0, st1.ms_total / 1e3, st1.projected), flush=True)
    assert st1.levels == 5 and st1.projected == int(cnt1.sum()) > 500_000 * n_windows
    sub = slice(0, 1500)
    g.set_option("chunk_ranges", 333)
    st2, cnt2, ck2 = g.query_batch_stats(ranges[sub], p5)
    assert (cnt2 == cnt1[sub]).all() and (ck2 == ck1[sub]).all()
    st3, cnt3, _ = g.query_batch_stats(ranges[sub], impg_amd.make_params(transitive=True, max_depth=3))
    assert (cnt3 <= cnt1[sub]).all() and int(cnt3.sum()) * 20 < int(cnt1[sub].sum())
