"""GPU parity: the HIP engine, called through the C ABI, against the CPU oracle
on the same inputs.  Bit-exact (integer work): same intervals, same order."""
import numpy as np
import pytest

import impg_amd
from impg_amd import index as gi
from oracle import oracle as o
from tests.paf_gen import random_paf, random_ranges

pytestmark = pytest.mark.gpu


def both(tmp_path, text, order=impg_amd.ORDER_COITREES, bidirectional=True):
    path = str(tmp_path / "t.paf")
    with open(path, "w") as f:
        f.write(text)
    g = impg_amd.GpuImpg.from_paf(path, bidirectional=bidirectional, order=order)
    c = o.OracleIndex(paf_paths=[path], bidirectional=bidirectional, preparse=True)
    assert g.num_seqs() == c.num_seqs()
    for i in range(g.num_seqs()):
        assert g.seq_name(i) == c.seq_name(i) and g.seq_len(i) == c.seq_len(i)
    return g, c


def assert_same(g, c, ranges, masked_regions=None, subset_keep=None, **kw):
    res = g.query_batch(ranges, impg_amd.make_params(**kw), masked_regions=masked_regions, subset_keep=subset_keep)
    total = 0
    for i, (t, s, e) in enumerate(ranges):
        want = c.query(t, s, e, masked_regions=masked_regions, subset_keep=subset_keep, **kw)
        got = res[i]
        assert got.tolist() == want.tolist(), (i, (t, s, e), kw)
        total += c.last_projection_count()
    assert res.projected == total
    return res


# ---- the reference's own known-answer vectors, through the C ABI ---------------
KATS = [  # (record ts,te,qs,qe,strand, cigar, range, expected (qs,qe,ts,te) or None)   impg.rs:2981-3156
    ((100, 200, 0, 100, "+"), "100=", (100, 200), (0, 100, 100, 200)),
    ((100, 200, 0, 100, "-"), "100=", (100, 200), (100, 0, 100, 200)),
    ((0, 100, 50, 200, "+"), "10=5I5D50=50I35=", (0, 100), (50, 200, 0, 100)),
    ((0, 100, 50, 200, "+"), "10=5I5D50=50I35=", (50, 55), (100, 105, 50, 55)),
    ((0, 100, 50, 200, "+"), "10=5I5D50=50I35=", (50, 64), (100, 114, 50, 64)),
    ((0, 100, 50, 200, "+"), "10=5I5D50=50I35=", (50, 65), (100, 165, 50, 65)),
    ((0, 100, 50, 200, "+"), "10=5I5D50=50I35=", (50, 66), (100, 166, 50, 66)),
    ((0, 100, 50, 200, "+"), "10=5I5D50=50I35=", (70, 95), (170, 195, 70, 95)),
    ((100, 200, 100, 200, "+"), "100=", (100, 200), (100, 200, 100, 200)),
    ((100, 200, 100, 200, "-"), "100=", (100, 200), (200, 100, 100, 200)),
    ((50, 150, 50, 160, "+"), "50=10I50=", (50, 150), (50, 160, 50, 150)),
    ((50, 150, 50, 140, "+"), "50=10D40=", (50, 150), (50, 140, 50, 150)),
    ((100, 200, 200, 300, "-"), "50=10D10I40=", (150, 250), (250, 200, 150, 200)),
    ((0, 50, 0, 40, "+"), "10=20D8=1X1=10I10=", (0, 10), (0, 10, 0, 10)),
]


@pytest.mark.parametrize("k", range(len(KATS)))
def test_reference_kats(tmp_path, k):
    (ts, te, qs, qe, strand), cg, rng, expect = KATS[k]
    text = "Q\t1000\t%d\t%d\t%s\tT\t1000\t%d\t%d\t1\t1\t60\tcg:Z:%s\n" % (qs, qe, strand, ts, te, cg)
    g, c = both(tmp_path, text, bidirectional=False)
    tid = g.seq_id("T")
    got = g.query(tid, rng[0], rng[1])
    assert got[0].tolist() == (tid, rng[0], rng[1], tid, rng[0], rng[1])
    assert got[1].tolist() == (g.seq_id("Q"), expect[0], expect[1], tid, expect[2], expect[3])
    assert got.tolist() == c.query(tid, rng[0], rng[1]).tolist()


def test_kat_65_65_not_emitted_and_touching(tmp_path):  # impg.rs:3029-3033
    text = "Q\t1000\t50\t200\t+\tT\t1000\t0\t100\t1\t1\t60\tcg:Z:10=5I5D50=50I35=\n"
    g, c = both(tmp_path, text, bidirectional=False)
    tid = g.seq_id("T")
    for rng in [(65, 66), (64, 65), (100, 120), (99, 100), (0, 1), (10, 15), (15, 16)]:
        assert g.query(tid, *rng).tolist() == c.query(tid, *rng).tolist()


# ---- tests/test_transitive_integrity.rs scenarios: BED bytes --------------------
L100 = "\t100\t100\t60\tcg:Z:100="
SCEN = [
    (["A\t1000\t0\t100\t+\tB\t1000\t0\t100" + L100, "A\t1000\t500\t600\t+\tC\t1000\t0\t100" + L100],
     ["A:0-100", "A:500-600"], dict(transitive=True)),
    (["A\t1000\t0\t100\t+\tB\t1000\t0\t100" + L100, "B\t1000\t0\t100\t+\tC\t1000\t0\t100" + L100],
     ["A:25-75"], dict(transitive=True)),
    (["A\t1000\t0\t100\t+\tB\t1000\t200\t300" + L100], ["A:0-100", "B:200-300"], dict()),
    (["A\t1000\t0\t100\t-\tB\t1000\t0\t100" + L100], ["A:0-50"], dict()),
    (["A\t2000\t0\t100\t+\tB\t1000\t0\t100" + L100, "A\t2000\t1000\t1100\t+\tC\t1000\t0\t100" + L100,
      "B\t1000\t0\t100\t+\tD\t1000\t0\t100" + L100, "C\t1000\t0\t100\t+\tD\t1000\t500\t600" + L100],
     ["A:0-100", "A:1000-1100"], dict(transitive=True, max_depth=3)),
    (["A\t1000\t0\t110\t+\tB\t1000\t0\t100\t100\t110\t60\tcg:Z:50=10I50="], ["A:0-50", "A:60-110"], dict()),
    (["A\t1000\t0\t100\t+\tB\t1000\t0\t100" + L100, "A\t1000\t0\t100\t+\tB\t1000\t500\t600" + L100], ["A:0-100"], dict()),
    (["A\t1000\t0\t100\t+\tB\t1000\t0\t100" + L100], ["A:500-600"], dict()),
    (["A\t1000\t0\t100\t+\tB\t1000\t0\t100" + L100, "B\t1000\t0\t100\t+\tC\t1000\t0\t100" + L100,
      "C\t1000\t0\t100\t+\tD\t1000\t0\t100" + L100], ["A:0-100"], dict(transitive=True, max_depth=1)),
    (["A\t1000\t0\t100\t+\tB\t1000\t0\t100" + L100, "B\t1000\t0\t100\t+\tC\t1000\t0\t100" + L100,
      "C\t1000\t0\t100\t+\tD\t1000\t0\t100" + L100], ["A:0-100"], dict(transitive=True, max_depth=2)),
]


@pytest.mark.parametrize("k", range(len(SCEN)))
def test_cli_scenarios_bed_bytes(tmp_path, k):
    lines, queries, kw = SCEN[k]
    kw = dict(kw, min_transitive_len=0)
    g, c = both(tmp_path, "\n".join(lines) + "\n")
    p = impg_amd.make_params(**kw)
    ranges, names = [], []
    for q in queries:
        name, s, e = gi.parse_target_range(q)
        ranges.append((g.seq_id(name), s, e))
        names.append(q)
    res = g.query_batch(ranges, p)
    got = res.bed(names, merge_distance=0, params=p)
    want = "".join(c.query_bed(gi.parse_target_range(q)[0], *gi.parse_target_range(q)[1:], range_name=q,
                               merge_distance=0, **kw) for q in queries)
    assert got == want
    assert got  # never empty: the self interval is always printed


# ---- randomised parity -----------------------------------------------------------
@pytest.mark.parametrize("seed,weird,incons", [(1, False, False), (2, True, False), (3, True, True), (4, False, True),
                                               (5, True, True)])
def test_random_query(tmp_path, seed, weird, incons):
    text, names = random_paf(seed, 400, n_seq=5, seq_len=30000, weird=weird, inconsistent=incons, self_aln=True)
    g, c = both(tmp_path, text)
    ranges = random_ranges(seed + 100, 300, 5, 30000, max_len=4000)
    assert_same(g, c, ranges)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_transitive(tmp_path, seed):
    text, names = random_paf(seed, 300, n_seq=6, seq_len=20000, weird=(seed % 2 == 0), self_aln=True)
    g, c = both(tmp_path, text)
    ranges = random_ranges(seed + 7, 60, 6, 20000, max_len=3000, min_len=50)
    for kw in [dict(transitive=True, max_depth=1, min_transitive_len=0, min_distance_between_ranges=0),
               dict(transitive=True, max_depth=2),
               dict(transitive=True, max_depth=3, min_transitive_len=20, min_distance_between_ranges=10),
               dict(transitive=True, max_depth=4, min_transitive_len=101, min_distance_between_ranges=50, min_output_length=200),
               dict(transitive=True, max_depth=0, min_transitive_len=300, min_distance_between_ranges=10)]:
        assert_same(g, c, ranges, **kw)


def test_dense_windows_many_hits(tmp_path):
    """> 64 candidates per range: the multi-chunk ordering path of lookup_emit."""
    rng = np.random.default_rng(5)
    lines = []
    for i in range(700):
        ts = int(rng.integers(0, 3000))
        ln = int(rng.integers(200, 2500))
        qs = int(rng.integers(0, 90000))
        lines.append("Q%d\t100000\t%d\t%d\t%s\tT\t10000\t%d\t%d\t1\t1\t60\tcg:Z:%d=" %
                     (i % 7, qs, qs + ln, "+-"[i % 2], ts, ts + ln, ln))
    g, c = both(tmp_path, "\n".join(lines) + "\n")
    tid = g.seq_id("T")
    ranges = [(tid, 1000, 2000), (tid, 0, 10000), (tid, 2999, 3001), (tid, 5000, 5400)]
    res = assert_same(g, c, ranges)
    assert len(res[1]) > 600
    assert_same(g, c, ranges, transitive=True, max_depth=2, min_transitive_len=50)


def test_sorted_order_policy(tmp_path):
    """IMPG_ORDER_SORTED: hits of a range come out in ascending target start."""
    text, names = random_paf(21, 300, n_seq=3, seq_len=20000)
    g, c = both(tmp_path, text, order=impg_amd.ORDER_SORTED)
    ranges = random_ranges(3, 50, 3, 20000, max_len=5000)
    res = g.query_batch(ranges, impg_amd.make_params())
    for i, (t, s, e) in enumerate(ranges):
        want = c.query(t, s, e)
        got = res[i]
        assert sorted(got.tolist()) == sorted(want.tolist())  # same set as the coitrees order
        assert got[0].tolist() == want[0].tolist()


def test_unknown_and_empty_targets(tmp_path):
    g, c = both(tmp_path, "A\t1000\t0\t100\t+\tB\t1000\t0\t100" + L100 + "\nC\t500\t0\t50\t+\tC\t500\t100\t150\t1\t1\t60\tcg:Z:50=\n")
    res = g.query_batch([(7, 0, 10), (0, 900, 950), (2, 0, 500)], impg_amd.make_params())
    assert res[0].tolist() == [(7, 0, 10, 7, 0, 10)]          # unknown target: self only (impg.rs:1896)
    assert res[1].tolist() == [(0, 900, 950, 0, 900, 950)]
    assert res[2].tolist() == c.query(2, 0, 500).tolist()      # self-alignment: forward entry only (impg.rs:1584)
    with pytest.raises(impg_amd.ImpgGpuError):
        g.query_batch([(0, 10, 10)], impg_amd.make_params())


def test_missing_cigar_is_an_error(tmp_path):
    g, c = both(tmp_path, "A\t1000\t0\t100\t+\tB\t1000\t0\t100\t1\t1\t60\n")
    assert g.query(0, 500, 600).tolist() == [(0, 500, 600, 0, 500, 600)]
    with pytest.raises(impg_amd.ImpgGpuError, match="CIGAR"):  # the reference panics (impg.rs:506-511)
        g.query(1, 0, 100)


def test_synthetic_config_small(tmp_path):
    """BASELINE configs 2/3 at reduced size: same generator, same flags."""
    path = str(tmp_path / "s.paf")
    shape = dict(n_seq=8, seq_len=400000, target_span=10000, n_blocks=100)
    impg_amd.synth_paf_text(path, 42, 1500, **shape)
    g = impg_amd.GpuImpg.from_paf(path)
    c = o.OracleIndex(paf_paths=[path], preparse=True)
    bed = impg_amd.synth_bed(7, 40, n_seq=8, seq_len=400000, range_len=5000)
    ranges = [(g.seq_id(impg_amd.synth_seq_name(int(r["target_id"]))), int(r["start"]), int(r["end"])) for r in bed]
    assert_same(g, c, ranges)
    assert_same(g, c, ranges, transitive=True, max_depth=3)
    p = impg_amd.make_params(transitive=True, max_depth=3)
    res = g.query_batch(ranges, p)
    got = res.bed(None, merge_distance=1000, params=p)
    want = "".join(c.query_bed(g.seq_name(t), s, e, merge_distance=1000, transitive=True, max_depth=3) for t, s, e in ranges)
    assert got == want
    # stats form agrees with the full form
    st, cnt, ck = g.query_batch_stats(ranges, p)
    assert st.projected == res.projected
    assert cnt.tolist() == [len(res[i]) - 1 for i in range(len(ranges))]


@pytest.mark.parametrize("n", [64, 65, 128, 129, 4096, 4097, 10000])
def test_large_segments_search_edges(tmp_path, n):
    """Segments larger than one wave: multi-round 64-ary search, including
    ranges before/after every entry (the 'no probe matches' round)."""
    L = 20 * n + 1000
    lines = ["q%d\t%d\t%d\t%d\t+\tT\t%d\t%d\t%d\t5\t5\t60\tcg:Z:12=" % (i % 3, L, 20 * i, 20 * i + 12, L, 20 * i + 100, 20 * i + 112)
             for i in range(n)]
    g, c = both(tmp_path, "\n".join(lines) + "\n", bidirectional=False)
    tid = g.seq_id("T")
    rng = np.random.default_rng(n)
    ranges = [(tid, 0, 50), (tid, 0, 100), (tid, 0, 101), (tid, L - 50, L), (tid, 20 * n + 92, L), (tid, 20 * n + 91, L),
              (tid, 0, L), (tid, 99, 101), (tid, 111, 113), (tid, 112, 120)]
    for _ in range(40):
        s = int(rng.integers(0, L - 400))
        ranges.append((tid, s, s + int(rng.integers(1, 400))))
    assert_same(g, c, ranges)
    assert_same(g, c, ranges, transitive=True, max_depth=1, min_transitive_len=0)


@pytest.mark.parametrize("seed,weird", [(31, False), (32, True), (33, True)])
def test_long_cigars_external_checkpoints(tmp_path, seed, weird):
    """Records with more than 8 tiles (> 224 ops): checkpoints live in the external array."""
    text, names = random_paf(seed, 160, n_seq=4, seq_len=400000, max_ops=3000, weird=weird, inconsistent=(seed == 33),
                             self_aln=True)
    g, c = both(tmp_path, text)
    ranges = random_ranges(seed, 150, 4, 400000, max_len=30000, min_len=1)
    assert_same(g, c, ranges)
    assert_same(g, c, ranges[:40], transitive=True, max_depth=2, min_transitive_len=50)


@pytest.mark.parametrize("seed,weird,max_ops", [(41, False, 150), (42, True, 150), (43, False, 1200)])
def test_min_gap_compressed_identity(tmp_path, seed, weird, max_ops):
    """--min-result-identity: calculate_gap_compressed_identity on the projected slice (impg.rs:1283-1287)."""
    text, names = random_paf(seed, 300, n_seq=5, seq_len=120000 if max_ops > 200 else 30000, max_ops=max_ops, weird=weird,
                             self_aln=True)
    g, c = both(tmp_path, text)
    sl = 120000 if max_ops > 200 else 30000
    ranges = random_ranges(seed, 200, 5, sl, max_len=6000)
    kept = []
    for thr in [0.0, 0.35, 0.5, 0.62, 0.8, 0.999, 1.0]:
        res = assert_same(g, c, ranges, min_identity=thr)
        kept.append(res.projected)
    assert kept[0] > kept[-1] and sorted(kept, reverse=True) == kept  # the filter bites, monotonically
    for thr in [0.4, 0.7]:
        assert_same(g, c, ranges[:50], transitive=True, max_depth=3, min_transitive_len=30, min_identity=thr)


@pytest.mark.parametrize("n", [150, 380, 470, 560, 700, 950])
def test_visit_order_network_widths(tmp_path, n):
    """lookup_emit_lane sorts a window's hits into visit order with a network pruned to the wave's widest window (32 / 40 /
    48 / 64 places, kernels.hip emit_sort): coverage swept so that the windows pass through every width and into the
    wave-per-range kernel beyond 64."""
    text, names = random_paf(300 + n, n, n_seq=2, seq_len=40000, max_ops=120, self_aln=True)
    g, c = both(tmp_path, text)
    ranges = random_ranges(n, 300, 2, 40000, max_len=1000)
    assert_same(g, c, ranges)
    assert_same(g, c, ranges[:80], transitive=True, max_depth=2, min_transitive_len=30)


def test_identity_threshold_on_the_border(tmp_path):
    """Thresholds equal to a hit's own identity and its two neighbouring doubles: the kernel decides most pairs by a
    comparison with slack and only borderline ones by the division (kernels.hip, project_pair) -- these are the borderline ones."""
    text, names = random_paf(44, 200, n_seq=4, seq_len=30000, max_ops=150, self_aln=True)
    g, c = both(tmp_path, text)
    ranges = random_ranges(44, 40, 4, 30000, max_len=5000)
    idents = set()
    for (t, s, e) in ranges:
        _, cg = c.query_cigar(t, s, e)
        idents.update(o.gap_compressed_identity(ops) for ops in cg if len(ops))
    idents = sorted(v for v in idents if 0.0 < v < 1.0)
    assert len(idents) > 20
    picks = idents[:: max(1, len(idents) // 25)]
    for v in picks:
        for thr in (v, float(np.nextafter(v, 1.0)), float(np.nextafter(v, 0.0))):
            assert_same(g, c, ranges, min_identity=thr)
    for thr in (1 / 3, 2 / 3, 0.1 + 0.2):
        assert_same(g, c, ranges, transitive=True, max_depth=2, min_transitive_len=30, min_identity=thr)


@pytest.mark.parametrize("seed", [51, 52, 53])
def test_transitive_dfs(tmp_path, seed):
    """query_transitive_dfs (impg.rs:2057-2309): one stack pop per query per round."""
    text, names = random_paf(seed, 300, n_seq=6, seq_len=20000, weird=(seed % 2 == 0), self_aln=True)
    g, c = both(tmp_path, text)
    ranges = random_ranges(seed + 7, 50, 6, 20000, max_len=3000, min_len=50)
    for kw in [dict(transitive=True, dfs=True, max_depth=1, min_transitive_len=0, min_distance_between_ranges=0),
               dict(transitive=True, dfs=True, max_depth=2),
               dict(transitive=True, dfs=True, max_depth=3, min_transitive_len=20, min_distance_between_ranges=10),
               dict(transitive=True, dfs=True, max_depth=4, min_transitive_len=101, min_distance_between_ranges=50,
                    min_output_length=200),
               dict(transitive=True, dfs=True, max_depth=0, min_transitive_len=300, min_distance_between_ranges=10)]:
        assert_same(g, c, ranges, **kw)


def test_deep_bfs_compacts_visited_tables(tmp_path):
    """A chain of 70 sequences: unlimited depth walks 69 levels, more than the
    visited-table limit, so the tables are folded on the way."""
    n = 70
    lines = ["S%d\t1000\t0\t500\t+\tS%d\t1000\t0\t500\t500\t500\t60\tcg:Z:500=" % (i, i + 1) for i in range(n - 1)]
    g, c = both(tmp_path, "\n".join(lines) + "\n")
    ranges = [(g.seq_id("S0"), 100, 400), (g.seq_id("S35"), 0, 500), (g.seq_id("S69"), 200, 450)]
    for kw in [dict(transitive=True, max_depth=0), dict(transitive=True, dfs=True, max_depth=0),
               dict(transitive=True, max_depth=60), dict(transitive=True, dfs=True, max_depth=65)]:
        res = assert_same(g, c, ranges, **kw)
    assert len(res[0]) >= 60


def test_cli_bed_bytes(tmp_path):
    """`impg-gpu query` prints the bytes the reference's `impg query -o bed` would (oracle restatement)."""
    import os
    import subprocess
    cli = os.path.join(os.path.dirname(impg_amd.__file__), "impg-gpu")
    text, names = random_paf(61, 300, n_seq=5, seq_len=30000, self_aln=True)
    paf = str(tmp_path / "c.paf")
    with open(paf, "w") as f:
        f.write(text)
    c = o.OracleIndex(paf_paths=[paf], preparse=True)
    rl = random_ranges(9, 25, 5, 30000, max_len=4000, min_len=150)
    bed = str(tmp_path / "q.bed")
    with open(bed, "w") as f:
        for i, (t, s, e) in enumerate(rl):
            f.write("%s\t%d\t%d%s\n" % (c.seq_name(t), s, e, "" if i % 3 == 0 else ("\tname%d" % i if i % 3 == 1 else "\t.")))
    rnames = [("%s:%d-%d" % (c.seq_name(t), s, e)) if i % 3 != 1 else "name%d" % i for i, (t, s, e) in enumerate(rl)]
    cases = [(["-d", "1000"], dict(), 1000), (["-d", "0", "-x", "-m", "3"], dict(transitive=True, max_depth=3), 0),
             (["--no-merge", "-x"], dict(transitive=True), -1),
             (["-d", "5k", "--transitive-dfs", "-m", "2", "-l", "300"], dict(transitive=True, dfs=True, max_depth=2, min_output_length=300), 5000),
             (["-d", "10", "--min-result-identity", "0.6", "-l", "500"], dict(min_identity=0.6, min_output_length=500), 10)]
    for flags, kw, d in cases:
        r = subprocess.run([cli, "query", "-a", paf, "-b", bed, "-o", "bed"] + flags, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        want = "".join(c.query_bed(c.seq_name(t), s, e, range_name=rnames[i], merge_distance=d, **kw) for i, (t, s, e) in enumerate(rl))
        assert r.stdout == want, flags
    # parse_merge_distance vectors (main.rs:13701-13707): metric suffixes, fractions
    t, s, e = rl[0]
    one = ["-r", "%s:%d-%d" % (c.seq_name(t), s, e)]
    for spelled, d in (("50000", 50000), ("50k", 50000), ("1m", 1000000), ("1M", 1000000), ("1.5k", 1500)):
        r = subprocess.run([cli, "query", "-a", paf] + one + ["-d", spelled], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout == c.query_bed(c.seq_name(t), s, e, merge_distance=d), spelled
    # -r form and the validation errors of main.rs:10387-10520
    t, s, e = rl[0]
    r = subprocess.run([cli, "query", "-a", paf, "-r", "%s:%d-%d" % (c.seq_name(t), s, e), "-d", "100"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == c.query_bed(c.seq_name(t), s, e, merge_distance=100)
    for bad in (["-r", "nope:1-500", "-d", "0"], ["-r", "%s:0-50" % c.seq_name(0), "-d", "0"], ["-r", "%s:0-500" % c.seq_name(0)],
                ["-r", "%s:0-99999999" % c.seq_name(0), "-d", "0"], ["-b", bed, "-d", "0", "-o", "gfa"]):
        r = subprocess.run([cli, "query", "-a", paf] + bad, capture_output=True, text=True)
        assert r.returncode != 0 and r.stdout == "" and r.stderr.startswith("Error:")
    # -o auto is BEDPE for a BED of targets (main.rs:7365-7373); -o paf
    for fmt_flags, fmt in ([], "bedpe"), (["-o", "paf"], "paf"), (["-o", "bedpe", "-x", "-m", "2"], "bedpe"):
        kw = dict(transitive=True, max_depth=2) if "-x" in fmt_flags else dict()
        r = subprocess.run([cli, "query", "-a", paf, "-b", bed, "-d", "50"] + fmt_flags, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        want = "".join(c.query_paf(c.seq_name(t), s, e, range_name=rnames[i], merge_distance=50, fmt=fmt, **kw)
                       for i, (t, s, e) in enumerate(rl))
        assert r.stdout == want, fmt_flags


@pytest.mark.parametrize("seed,weird,incons,max_ops", [(71, False, False, 150), (72, True, False, 150), (73, True, True, 150),
                                                        (74, False, False, 1500)])
def test_store_cigar_slices(tmp_path, seed, weird, incons, max_ops):
    """store_cigar = true: every interval's Vec<CigarOp> (sliced, first/last op trimmed, inverted for
    reversed entries -- impg.rs:2878-2886, :144-156) equals the oracle's, op for op."""
    sl = 200000 if max_ops > 200 else 30000
    text, names = random_paf(seed, 250, n_seq=5, seq_len=sl, max_ops=max_ops, weird=weird, inconsistent=incons, self_aln=True)
    g, c = both(tmp_path, text)
    ranges = random_ranges(seed, 120, 5, sl, max_len=8000)
    for kw in [dict(), dict(transitive=True, max_depth=2, min_transitive_len=40), dict(min_identity=0.5),
               dict(transitive=True, dfs=True, max_depth=2, min_transitive_len=40)]:
        rs = ranges if not kw.get("transitive") else ranges[:40]
        res = g.query_batch(rs, impg_amd.make_params(store_cigar=True, **kw))
        for i, (t, s, e) in enumerate(rs):
            want, wcg = c.query_cigar(t, s, e, **kw)
            assert res[i].tolist() == want.tolist(), (i, kw)
            got = res.cigars(i)
            assert len(got) == len(wcg)
            for k in range(len(wcg)):
                assert got[k].tolist() == wcg[k].tolist(), (i, k, kw)
    # the reference's own slice vectors (impg.rs:3016-3053) come out of the same path
    text = "Q\t1000\t50\t200\t+\tT\t1000\t0\t100\t1\t1\t60\tcg:Z:10=5I5D50=50I35=\n"
    g, c = both(tmp_path, text, bidirectional=False)
    tid = g.seq_id("T")
    r = g.query_batch([(tid, 50, 65), (tid, 50, 66), (tid, 0, 100)], impg_amd.make_params(store_cigar=True))
    assert o.ops_to_pairs(r.cigars(0)[1]) == [(15, "="), (50, "I")]
    assert o.ops_to_pairs(r.cigars(1)[1]) == [(15, "="), (50, "I"), (1, "=")]
    assert o.ops_to_pairs(r.cigars(2)[1]) == [(10, "="), (5, "I"), (5, "D"), (50, "="), (50, "I"), (35, "=")]
    assert o.ops_to_pairs(r.cigars(2)[0]) == [(100, "=")]  # the self interval's own CIGAR (impg.rs:1870-1872)


def both_files(tmp_path, texts, bidirectional=True):
    paths = []
    for i, t in enumerate(texts):
        paths.append(str(tmp_path / ("f%d.paf" % i)))
        with open(paths[-1], "w") as f:
            f.write(t)
    g = impg_amd.GpuImpg.from_paf(paths, bidirectional=bidirectional)
    c = o.OracleIndex(paf_paths=paths, bidirectional=bidirectional, preparse=True)
    assert g.num_seqs() == c.num_seqs()
    return g, c


@pytest.mark.parametrize("seed,n_files", [(81, 1), (82, 3), (83, 4)])
def test_multi_impg_semantics(tmp_path, seed, n_files):
    """params.multi_impg: MultiImpg::query / query_transitive_{bfs,dfs} over per-file indices
    (multi_impg.rs:495-595, :796-991): hits of one step merged over the files and sorted by five
    keys, one worklist pop at a time (front or back), unclipped ranges, same-sequence hits skipped."""
    texts = []
    for k in range(n_files):  # same sequence universe in every file, different alignments
        t, names = random_paf(seed * 10 + k, 120, n_seq=6, seq_len=20000, weird=(k % 2 == 1), self_aln=True)
        texts.append(t)
    g, c = both_files(tmp_path, texts)
    ranges = random_ranges(seed + 3, 60, g.num_seqs(), 20000, max_len=3000, min_len=50)
    for kw in [dict(),
               dict(min_identity=0.6),
               dict(transitive=True, max_depth=1, min_transitive_len=0, min_distance_between_ranges=0),
               dict(transitive=True, max_depth=2),
               dict(transitive=True, dfs=True, max_depth=2),
               dict(transitive=True, max_depth=3, min_transitive_len=20, min_distance_between_ranges=10),
               dict(transitive=True, dfs=True, max_depth=4, min_transitive_len=101, min_distance_between_ranges=50,
                    min_output_length=200),
               dict(transitive=True, max_depth=0, min_transitive_len=300, min_distance_between_ranges=10)]:
        assert_same(g, c, ranges, multi_impg=True, **kw)
    # store_cigar rides along with the permutation of the sorted hits
    kw = dict(transitive=True, max_depth=2, min_transitive_len=40, multi_impg=True)
    res = g.query_batch(ranges[:20], impg_amd.make_params(store_cigar=True, **kw))
    for i, (t, s, e) in enumerate(ranges[:20]):
        want, wcg = c.query_cigar(t, s, e, **kw)
        assert res[i].tolist() == want.tolist()
        got = res.cigars(i)
        assert [x.tolist() for x in got] == [x.tolist() for x in wcg]


def test_multi_impg_dense_step(tmp_path):
    """> 64 hits in one step: the multi-chunk path of the five-key sort, with duplicates
    of the self interval and exact ties."""
    lines = []
    for i in range(150):
        q = "Q%d" % (i % 7)
        lines.append("%s\t5000\t%d\t%d\t%s\tT\t5000\t%d\t%d\t10\t10\t60\tcg:Z:%d=" %
                     (q, 100 + (i % 5), 600 + (i % 5), "+-"[i % 2], 1000 + (i % 3), 1500 + (i % 3), 500))
    lines.append("T\t5000\t1000\t1500\t+\tT\t5000\t1000\t1500\t10\t10\t60\tcg:Z:500=")  # maps a range onto itself
    g, c = both_files(tmp_path, ["\n".join(lines[:80]) + "\n", "\n".join(lines[80:]) + "\n"])
    t = g.seq_id("T")
    ranges = [(t, 1000, 1500), (t, 900, 1600), (t, 1200, 1300), (g.seq_id("Q1"), 0, 1000)]
    for kw in [dict(), dict(transitive=True, max_depth=2, min_transitive_len=10),
               dict(transitive=True, dfs=True, max_depth=3, min_transitive_len=10, min_distance_between_ranges=0)]:
        assert_same(g, c, ranges, multi_impg=True, **kw)


@pytest.mark.parametrize("seed,weird,incons,max_ops", [(91, False, False, 60), (92, True, False, 60), (93, True, True, 200),
                                                        (94, False, False, 12)])
def test_paf_and_bedpe_bytes(tmp_path, seed, weird, incons, max_ops):
    """`-o paf` / `-o bedpe`: merge_adjusted_intervals (CIGAR concatenation, gap filling, f32-scaled trims)
    and the gi:f / bi:f columns, byte for byte against the oracle's restatement (main.rs:11894-12103,
    :12563-12845, :13014-13180)."""
    sl = 30000
    text, names = random_paf(seed, 300, n_seq=5, seq_len=sl, max_ops=max_ops, weird=weird, inconsistent=incons, self_aln=True)
    g, c = both(tmp_path, text)
    ranges = random_ranges(seed, 60, 5, sl, max_len=6000, min_len=150)
    rnames = ["r%d" % i for i in range(len(ranges))]
    for kw in [dict(), dict(transitive=True, max_depth=2), dict(transitive=True, dfs=True, max_depth=3, min_transitive_len=50),
               dict(min_identity=0.7), dict(min_output_length=300)]:
        p = impg_amd.make_params(store_cigar=True, **kw)
        res = g.query_batch(ranges, p)

        def oracle_text(i, d, fmt):
            t, s, e = ranges[i]
            return c.query_paf(g.seq_name(t), s, e, range_name=rnames[i], merge_distance=d, fmt=fmt, **kw)

        # a range whose every row is filtered away makes the reference panic in results.remove(0): both sides raise
        dead = set()
        for i in range(len(ranges)):
            try:
                oracle_text(i, 0, "paf")
            except RuntimeError:
                dead.add(i)
                one = g.query_batch([ranges[i]], p)
                with pytest.raises(impg_amd.ImpgGpuError):
                    one.paf([rnames[i]], merge_distance=0, params=p)
        live = [i for i in range(len(ranges)) if i not in dead]
        if dead:
            res = g.query_batch([ranges[i] for i in live], p)
        for d in (-1, 0, 25, 1000):
            for fmt in ("paf", "bedpe"):
                got = res.paf([rnames[i] for i in live], merge_distance=d, params=p, fmt=fmt)
                want = "".join(oracle_text(i, d, fmt) for i in live)
                assert got == want, (kw, d, fmt)
    # chained tandem copies: contiguous rows that merge_adjusted_intervals joins
    lines = ["Q\t4000\t%d\t%d\t+\tT\t4000\t%d\t%d\t1\t1\t60\tcg:Z:%d=" % (100 * k, 100 * k + 100, 500 + 100 * k, 600 + 100 * k, 100)
             for k in range(8)]
    lines += ["R\t4000\t%d\t%d\t-\tT\t4000\t%d\t%d\t1\t1\t60\tcg:Z:40=3I57=" % (3000 - 110 * k, 3100 - 110 * k, 500 + 100 * k, 600 + 100 * k)
              for k in range(6)]
    g, c = both(tmp_path, "\n".join(lines) + "\n", bidirectional=False)
    t = g.seq_id("T")
    p = impg_amd.make_params(store_cigar=True)
    res = g.query_batch([(t, 400, 1500)], p)
    for d in (0, 5, 20):
        for fmt in ("paf", "bedpe"):
            got = res.paf(["x"], merge_distance=d, params=p, fmt=fmt)
            assert got == c.query_paf("T", 400, 1500, range_name="x", merge_distance=d, fmt=fmt)
    assert res.paf(["x"], merge_distance=0, params=p).count("\n") < res.paf(["x"], merge_distance=-1, params=p).count("\n")


def test_projection_order_is_invisible(tmp_path):
    """locality_min = 1 forces the window-order projection (slot_of indirection, XCD block mapping, lane
    and wave emit passes writing it) on small batches: results must not change in any position."""
    text, names = random_paf(131, 400, n_seq=6, seq_len=20000, weird=True, self_aln=True)
    g, c = both(tmp_path, text)
    g.set_option("locality_min", 1)
    ranges = random_ranges(17, 80, 6, 20000, max_len=4000, min_len=50)
    for kw in [dict(), dict(transitive=True, max_depth=3, min_transitive_len=20), dict(transitive=True, dfs=True, max_depth=2),
               dict(min_identity=0.6), dict(transitive=True, max_depth=2, multi_impg=True)]:
        assert_same(g, c, ranges, **kw)
    res = g.query_batch(ranges[:30], impg_amd.make_params(store_cigar=True, transitive=True, max_depth=2))
    for i, (t, s, e) in enumerate(ranges[:30]):
        want, wcg = c.query_cigar(t, s, e, transitive=True, max_depth=2)
        assert res[i].tolist() == want.tolist()
        assert [x.tolist() for x in res.cigars(i)] == [x.tolist() for x in wcg]
    # dense target: windows wider than 64 entries go through the wave-per-range emit pass
    lines = []
    for i in range(300):
        lines.append("Q%d\t9000\t%d\t%d\t%s\tT\t9000\t%d\t%d\t10\t10\t60\tcg:Z:%d=" %
                     (i % 11, 100 + i, 1100 + i, "+-"[i % 2], 2000 + (i * 7) % 900, 3000 + (i * 7) % 900, 1000))
    g, c = both(tmp_path, "\n".join(lines) + "\n")
    g.set_option("locality_min", 1)
    t = g.seq_id("T")
    dense = [(t, 2500, 2600), (t, 2000, 4000), (t, 2890, 2910), (g.seq_id("Q3"), 0, 2000)]
    for kw in [dict(), dict(transitive=True, max_depth=2, min_transitive_len=10)]:
        assert_same(g, c, dense, **kw)


def test_many_small_targets(tmp_path):
    """Hundreds of sequences with a handful of alignments each (tiny segments, most without a sampled level,
    some empty), lookup order on: key arithmetic, segment tables and the (query, sequence) keys at another scale."""
    text, names = random_paf(151, 1500, n_seq=400, seq_len=3000, self_aln=True)
    g, c = both(tmp_path, text)
    g.set_option("locality_min", 1)
    ranges = random_ranges(23, 300, g.num_seqs(), 3000, max_len=1500, min_len=30)
    for kw in [dict(), dict(transitive=True, max_depth=3, min_transitive_len=10, min_distance_between_ranges=0),
               dict(transitive=True, dfs=True, max_depth=2, min_transitive_len=10),
               dict(transitive=True, max_depth=0, min_transitive_len=400)]:
        assert_same(g, c, ranges, **kw)


def test_multi_impg_ties_keep_file_order(tmp_path):
    """Hits that agree on all five MultiImpg sort keys but carry different CIGARs (the same alignment
    coordinates in two files): the stable sort of multi_impg.rs:582-592 keeps them file by file, each in its own
    tree's visit order -- found by scripts/fuzz_parity.py."""
    def rec(q, qs, qe, t, ts, te, cg):
        return "%s\t5000\t%d\t%d\t+\t%s\t5000\t%d\t%d\t1\t1\t60\tcg:Z:%s" % (q, qs, qe, t, ts, te, cg)
    f0 = [rec("A", 100, 200, "T", 1000, 1100, "100="), rec("B", 0, 50, "T", 1020, 1070, "50="),
          rec("A", 100, 200, "T", 1000, 1100, "40=1X59="), rec("C", 10, 110, "T", 900, 1000, "100=")]
    f1 = [rec("A", 100, 200, "T", 1000, 1100, "10=2X88="), rec("A", 100, 200, "T", 1000, 1100, "99=1X"),
          rec("D", 5, 105, "T", 1050, 1150, "100=")] + [rec("E%d" % k, 0, 30, "T", 1000 + k, 1030 + k, "30=") for k in range(12)]
    g, c = both_files(tmp_path, ["\n".join(f0) + "\n", "\n".join(f1) + "\n"])
    t = g.seq_id("T")
    ranges = [(t, 1000, 1100), (t, 950, 1200), (t, 1040, 1060)]
    for kw in [dict(multi_impg=True), dict(multi_impg=True, transitive=True, max_depth=2, min_transitive_len=10)]:
        res = g.query_batch(ranges, impg_amd.make_params(store_cigar=True, **kw))
        for i, (tt, s, e) in enumerate(ranges):
            want, wcg = c.query_cigar(tt, s, e, **kw)
            assert res[i].tolist() == want.tolist()
            assert [x.tolist() for x in res.cigars(i)] == [x.tolist() for x in wcg], (i, kw)


def test_gzip_and_bgzf_paf_ingest(tmp_path):
    """.paf.gz (one gzip stream) and BGZF-style input (a series of gzip members): the same index as from the text."""
    import gzip
    text, names = random_paf(171, 300, n_seq=5, seq_len=20000, self_aln=True)
    g, c = both(tmp_path, text)
    gz = str(tmp_path / "t.paf.gz")
    with gzip.open(gz, "wb") as f:
        f.write(text.encode())
    bgz = str(tmp_path / "t.paf.bgz")
    raw = text.encode()
    with open(bgz, "wb") as f:  # members cut at arbitrary byte positions, then the empty EOF member
        for a in range(0, len(raw), 7001):
            f.write(gzip.compress(raw[a:a + 7001]))
        f.write(gzip.compress(b""))
    ranges = random_ranges(5, 60, 5, 20000, max_len=3000, min_len=50)
    want = g.query_batch(ranges, impg_amd.make_params(transitive=True, max_depth=2))
    for path in (gz, bgz):
        g2 = impg_amd.GpuImpg.from_paf(path)
        assert g2.num_seqs() == g.num_seqs() and [g2.seq_name(i) for i in range(g.num_seqs())] == [g.seq_name(i) for i in range(g.num_seqs())]
        got = g2.query_batch(ranges, impg_amd.make_params(transitive=True, max_depth=2))
        assert all(got[i].tolist() == want[i].tolist() for i in range(len(ranges)))
    assert_same(g, c, ranges[:20], transitive=True, max_depth=2)


def random_mask(seed, n_seq, seq_len, present=0.8, max_ranges=12, odd_lengths=False):
    """{seq id: (sequence_length, sorted disjoint non-touching ranges)} -- a masked_regions map."""
    rng = np.random.default_rng(seed)
    mask = {}
    for sid in range(n_seq):
        if rng.random() > present:
            continue
        k = int(rng.integers(0, max_ranges + 1))
        cuts = np.unique(rng.integers(0, seq_len, size=2 * k))
        cuts = cuts[: 2 * (len(cuts) // 2)]
        rs = [(int(cuts[2 * i]), int(cuts[2 * i + 1])) for i in range(len(cuts) // 2)]
        length = seq_len
        if odd_lengths and rng.random() < 0.3:  # a SortedRanges may carry any sequence_length
            length = int(rng.integers(seq_len // 2, seq_len + 1000))
        mask[sid] = (length, rs)
    return mask


@pytest.mark.parametrize("seed,present,odd", [(101, 1.0, False), (102, 0.7, False), (103, 0.9, True), (104, 0.3, True)])
def test_masked_regions(tmp_path, seed, present, odd):
    """query_transitive_{bfs,dfs} with masked_regions = Some(map) (impg.rs:2077-2112, :2331-2373;
    partition.rs:364, :380): one map for the batch, every range starting from its own clone.  Several
    self intervals (or none) per range, visited sets seeded from the map, the length-0 sets of
    sequences the map does not hold."""
    text, names = random_paf(seed, 300, n_seq=6, seq_len=20000, weird=(seed % 2 == 0), self_aln=True)
    g, c = both(tmp_path, text)
    ranges = random_ranges(seed + 7, 80, 6, 20000, max_len=4000, min_len=50)
    for m_seed in (1, 2):
        mask = random_mask(seed * 10 + m_seed, 6, 20000, present=present, odd_lengths=odd)
        for kw in [dict(transitive=True, max_depth=1, min_transitive_len=0, min_distance_between_ranges=0),
                   dict(transitive=True, max_depth=2),
                   dict(transitive=True, dfs=True, max_depth=3, min_transitive_len=20, min_distance_between_ranges=10),
                   dict(transitive=True, max_depth=4, min_transitive_len=101, min_distance_between_ranges=50,
                        min_output_length=200),
                   dict(transitive=True, dfs=True, max_depth=0, min_transitive_len=300, min_distance_between_ranges=10),
                   dict(transitive=True, max_depth=0, min_transitive_len=50, min_distance_between_ranges=0)]:
            assert_same(g, c, ranges, masked_regions=mask, **kw)
    # the empty map and the all-covering map
    assert_same(g, c, ranges, masked_regions={}, transitive=True, max_depth=2)
    assert_same(g, c, ranges, masked_regions={i: (20000, [(0, 20000)]) for i in range(6)}, transitive=True, max_depth=2)
    # store_cigar: every self piece carries its own N= (impg.rs:2352-2354)
    mask = random_mask(seed, 6, 20000, present=1.0)
    kw = dict(transitive=True, max_depth=2, min_transitive_len=40)
    res = g.query_batch(ranges[:20], impg_amd.make_params(store_cigar=True, **kw), masked_regions=mask)
    plain = g.query_batch(ranges[:20], impg_amd.make_params(**kw), masked_regions=mask)
    n_self = 0
    for i, (t, s0, e0) in enumerate(ranges[:20]):
        assert res[i].tolist() == plain[i].tolist()
        want, wcg = c.query(t, s0, e0, masked_regions=mask, **kw), None
        assert res[i].tolist() == want.tolist()
        for k, row in enumerate(res[i].tolist()):  # the leading self pieces
            if not (row[0] == row[3] == t and row[1:3] == row[4:6]):
                break
            assert res.cigars(i)[k].tolist() == o.ops_from_pairs([(row[2] - row[1], "=")]).tolist()
            n_self += 1
    assert n_self > 0
    # only the transitive queries take a mask
    with pytest.raises(impg_amd.ImpgGpuError):
        g.query_batch(ranges[:2], impg_amd.make_params(transitive=False), masked_regions=mask)
    # the trait-shaped calls
    t, s, e = ranges[0]
    got = g.query_transitive_bfs(t, s, e, masked_regions=mask, max_depth=3)
    assert got.tolist() == c.query(t, s, e, masked_regions=mask, transitive=True, max_depth=3).tolist()
    got = g.query_transitive_dfs(t, s, e, masked_regions=mask, max_depth=3)
    assert got.tolist() == c.query(t, s, e, masked_regions=mask, transitive=True, dfs=True, max_depth=3).tolist()


@pytest.mark.parametrize("seed", [111, 112])
def test_masked_regions_multi_impg(tmp_path, seed):
    """MultiImpg with masked_regions (multi_impg.rs:814-830, :919-922): sequences absent from the map keep
    their real length, except the range's own target."""
    texts = []
    for k in range(3):
        t, names = random_paf(seed * 10 + k, 120, n_seq=6, seq_len=20000, weird=(k % 2 == 1), self_aln=True)
        texts.append(t)
    g, c = both_files(tmp_path, texts)
    ranges = random_ranges(seed + 3, 60, g.num_seqs(), 20000, max_len=3000, min_len=50)
    for m_seed, present in ((1, 1.0), (2, 0.6)):
        mask = random_mask(seed * 10 + m_seed, g.num_seqs(), 20000, present=present, odd_lengths=True)
        for kw in [dict(transitive=True, max_depth=2),
                   dict(transitive=True, dfs=True, max_depth=3, min_transitive_len=20, min_distance_between_ranges=10),
                   dict(transitive=True, max_depth=0, min_transitive_len=300, min_distance_between_ranges=10)]:
            assert_same(g, c, ranges, masked_regions=mask, multi_impg=True, **kw)


@pytest.mark.parametrize("n_files", [1, 3])
def test_saved_index_round_trip(tmp_path, n_files):
    """impg_gpu_index_save / _load (the role of the reference's .impg file, impg.rs:1655-1850): a loaded index answers
    every kind of query exactly like the one that was built from the alignments, names and lengths included; the
    CLI builds it with `index` and reads it back with `query -i` without the alignment files."""
    import os
    import subprocess
    texts = [random_paf(700 + k, 150, n_seq=6, seq_len=20000, weird=(k == 1), self_aln=True, max_ops=300)[0] for k in range(n_files)]
    g, c = both_files(tmp_path, texts)
    saved = str(tmp_path / "index.impghbm")
    g.save(saved)
    h = impg_amd.GpuImpg.load(saved)
    assert h.num_seqs() == g.num_seqs() and h.num_entries() == g.num_entries() and h.num_records() == g.num_records()
    assert h.num_targets() == g.num_targets() and h.target_ids().tolist() == g.target_ids().tolist()
    assert h.device_bytes() == g.device_bytes()
    for i in range(g.num_seqs()):
        assert h.seq_name(i) == g.seq_name(i) and h.seq_len(i) == g.seq_len(i) and h.seq_id(g.seq_name(i)) == i
    ranges = random_ranges(5, 50, g.num_seqs(), 20000, max_len=3000, min_len=50)
    for kw in [dict(), dict(min_identity=0.7), dict(transitive=True, max_depth=3, min_transitive_len=20),
               dict(transitive=True, dfs=True, max_depth=2), dict(transitive=True, max_depth=2, multi_impg=True)]:
        assert_same(h, c, ranges, **kw)
    kw = dict(transitive=True, max_depth=2, min_transitive_len=40)
    a = g.query_batch(ranges[:10], impg_amd.make_params(store_cigar=True, **kw))
    b = h.query_batch(ranges[:10], impg_amd.make_params(store_cigar=True, **kw))
    for i in range(10):
        assert a[i].tolist() == b[i].tolist() and [x.tolist() for x in a.cigars(i)] == [x.tolist() for x in b.cigars(i)]
    # an index sharded over GPUs is saved part by part (tests/test_multi_gpu.py::test_multi_handle_save_load); its parts are
    # not plain indexes
    paths0 = [str(tmp_path / ("f%d.paf" % k)) for k in range(n_files)]
    sh = impg_amd.GpuImpg.from_paf(paths0, devices=[0, 0, 0])
    sh.save(str(tmp_path / "sharded.impghbm"))
    del sh
    for part in ("sharded.impghbm", "sharded.impghbm.shard0of3"):
        with pytest.raises(impg_amd.ImpgGpuError):
            impg_amd.GpuImpg.load(str(tmp_path / part))
    # a second save of the loaded index is the same file (of a fresh load: `h` has built its identity lines for the filter
    # above, test_identity_lines_on_demand, and would save them too)
    again = str(tmp_path / "again.impghbm")
    impg_amd.GpuImpg.load(saved).save(again)
    assert open(saved, "rb").read() == open(again, "rb").read()
    h.save(again)
    assert len(open(again, "rb").read()) > len(open(saved, "rb").read())
    # damaged / foreign files are refused
    blob = open(saved, "rb").read()
    for bad in (blob[:len(blob) // 2], b"IMPGIDX2" + blob[8:], blob[:-8] + bytes(8)):
        p = str(tmp_path / "bad.bin")
        open(p, "wb").write(bad)
        with pytest.raises(impg_amd.ImpgGpuError):
            impg_amd.GpuImpg.load(p)
    # CLI: index once, query from the saved file alone
    cli = os.path.join(os.path.dirname(impg_amd.__file__), "impg-gpu")
    pafs = [str(tmp_path / ("f%d.paf" % k)) for k in range(n_files)]
    for p, t in zip(pafs, texts):
        open(p, "w").write(t)
    cli_saved = str(tmp_path / "cli.impghbm")
    r = subprocess.run([cli, "index", "-a"] + pafs + ["-i", cli_saved], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    t, s, e = next(x for x in ranges if x[2] - x[1] >= 150)
    rng = "%s:%d-%d" % (c.seq_name(t), s, e)
    r1 = subprocess.run([cli, "query", "-i", cli_saved, "-r", rng, "-d", "100", "-x"], capture_output=True, text=True)
    r2 = subprocess.run([cli, "query", "-a"] + pafs + ["-r", rng, "-d", "100", "-x"], capture_output=True, text=True)
    assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr, r2.stderr)
    assert r1.stdout == r2.stdout == c.query_bed(c.seq_name(t), s, e, merge_distance=100, transitive=True)


@pytest.mark.parametrize("seed", [121, 122, 123])
def test_subset_filter(tmp_path, seed):
    """subset_filter = Some(..) (impg.rs:2176-2185, :2430-2439; multi_impg.rs:888-896; main.rs:11693-11696) as the
    host's per-sequence verdict: dropped hits are neither reported nor expanded; the query's own target always
    stays; non-transitive queries are filtered after the fact; with and without a mask; store_cigar rides along."""
    texts = [random_paf(seed * 10 + k, 150, n_seq=8, seq_len=20000, weird=(k == 1), self_aln=True)[0] for k in range(2)]
    g, c = both_files(tmp_path, texts)
    n = g.num_seqs()
    ranges = random_ranges(seed + 3, 70, n, 20000, max_len=3000, min_len=50)
    rng = np.random.default_rng(seed)
    for frac in (0.0, 0.4, 0.8, 1.0):
        keep = (rng.random(n) < frac).astype(np.uint8)
        for kw in [dict(), dict(min_identity=0.6), dict(multi_impg=True),
                   dict(transitive=True, max_depth=3, min_transitive_len=20, min_distance_between_ranges=10),
                   dict(transitive=True, dfs=True, max_depth=0, min_transitive_len=101),
                   dict(transitive=True, max_depth=3, min_transitive_len=50, multi_impg=True),
                   dict(transitive=True, dfs=True, max_depth=2, multi_impg=True, min_output_length=100)]:
            assert_same(g, c, ranges, subset_keep=keep, **kw)
        mask = random_mask(seed * 7, n, 20000, present=0.9)
        assert_same(g, c, ranges, masked_regions=mask, subset_keep=keep, transitive=True, max_depth=3, min_transitive_len=30)
    keep = (rng.random(n) < 0.5).astype(np.uint8)
    kw = dict(transitive=True, max_depth=2, min_transitive_len=40)
    a = g.query_batch(ranges[:15], impg_amd.make_params(store_cigar=True, **kw), subset_keep=keep)
    full = g.query_batch(ranges[:15], impg_amd.make_params(store_cigar=True, **kw))
    n_dropped = 0
    for i, (t, s0, e0) in enumerate(ranges[:15]):
        assert a[i].tolist() == c.query(t, s0, e0, subset_keep=keep, **kw).tolist()
        # depth 0 hits survive the filter with the CIGARs they have without it
        rows_full = {tuple(r): k for k, r in enumerate(full[i].tolist())}
        for k, r in enumerate(a[i].tolist()):
            if tuple(r) in rows_full and r[3] == t:
                assert a.cigars(i)[k].tolist() == full.cigars(i)[rows_full[tuple(r)]].tolist()
        n_dropped += len(full[i]) - len(a[i])
    assert n_dropped > 0
    with pytest.raises(impg_amd.ImpgGpuError):
        g.query_batch(ranges[:2], impg_amd.make_params(), subset_keep=keep[:-1])
    t, s0, e0 = ranges[0]
    got = g.query_transitive_bfs(t, s0, e0, subset_filter=keep, max_depth=3)
    assert got.tolist() == c.query(t, s0, e0, subset_keep=keep, transitive=True, max_depth=3).tolist()


def test_cli_subset_sequence_list(tmp_path):
    """`impg-gpu query --subset-sequence-list`: the list is read and matched on the host (impg_gpu_subset_keep), the
    verdicts filter on the device; same bytes as the library calls, and the reference's two error messages."""
    import os
    import subprocess
    cli = os.path.join(os.path.dirname(impg_amd.__file__), "impg-gpu")
    text, names = random_paf(131, 250, n_seq=8, seq_len=20000, self_aln=True)
    g, c = both(tmp_path, text)
    paf = str(tmp_path / "t.paf")
    all_names = [g.seq_name(i) for i in range(g.num_seqs())]
    lst = str(tmp_path / "subset.txt")
    with open(lst, "w") as f:
        f.write("# kept\n%s\n  %s:5-10\t\n\n%s\n" % (all_names[1], all_names[3], all_names[6]))
    keep, entries = impg_amd.subset_keep(open(lst).read(), all_names)
    assert entries == 3 and keep.tolist() == [1 if i in (1, 3, 6) else 0 for i in range(len(all_names))]
    t, s, e = next(x for x in random_ranges(4, 40, 8, 20000, max_len=3000, min_len=200))
    rng = "%s:%d-%d" % (g.seq_name(t), s, e)
    for flags, kw in ((["-x", "-m", "3"], dict(transitive=True, max_depth=3)), ([], dict())):
        r = subprocess.run([cli, "query", "-a", paf, "-r", rng, "-d", "50", "--subset-sequence-list", lst] + flags,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        p = impg_amd.make_params(**kw)
        want = g.query_batch([(t, s, e)], p, subset_keep=keep).bed([rng], merge_distance=50, params=p)
        assert r.stdout == want
        assert g.query_batch([(t, s, e)], p, subset_keep=keep)[0].tolist() == c.query(t, s, e, subset_keep=keep, **kw).tolist()
        full = subprocess.run([cli, "query", "-a", paf, "-r", rng, "-d", "50"] + flags, capture_output=True, text=True)
        assert full.stdout != r.stdout  # (the list does change the answer)
    empty = str(tmp_path / "empty.txt")
    open(empty, "w").write("# nothing\n\n")
    for bad in (empty, str(tmp_path / "missing.txt")):
        r = subprocess.run([cli, "query", "-a", paf, "-r", rng, "-d", "50", "--subset-sequence-list", bad], capture_output=True, text=True)
        assert r.returncode != 0 and r.stdout == "" and r.stderr.startswith("Error:")


def test_original_sequence_coordinates(tmp_path):
    """--original-sequence-coordinates (main.rs:4370, :4642-4678): sequences named "base:START-END" are printed as
    "base" with START added to their coordinates in BED / BEDPE / PAF; PAF lengths become 0 without sequence files."""
    import os
    import subprocess
    text, names = random_paf(141, 200, n_seq=6, seq_len=20000, self_aln=True)
    # rename: s0 -> chrA:1000-21000, s1 -> sample#1#chrB:5-20005, s2 -> chrC:notanumber-5, the rest keep their names
    ren = {"s0": "chrA:1000-21000", "s1": "sample#1#chrB:5-20005", "s2": "chrC:oops-5"}
    lines = []
    for ln in text.splitlines():
        f = ln.split("\t")
        f[0], f[5] = ren.get(f[0], f[0]), ren.get(f[5], f[5])
        lines.append("\t".join(f))
    g, c = both(tmp_path, "\n".join(lines) + "\n")
    ranges = [(t, s, e) for (t, s, e) in random_ranges(9, 30, 6, 20000, max_len=3000, min_len=150)]
    rnames = ["r%d" % i for i in range(len(ranges))]
    kw = dict(transitive=True, max_depth=2, min_transitive_len=40)
    for orig in (True, False):
        p = impg_amd.make_params(original_sequence_coordinates=orig, **kw)
        want = "".join(c.query_bed(c.seq_name(t), s, e, range_name=rnames[i], merge_distance=20, original_sequence_coordinates=orig, **kw)
                       for i, (t, s, e) in enumerate(ranges))
        got = g.query_batch(ranges, p).bed(rnames, merge_distance=20, params=p)
        assert got == want
        pc = impg_amd.make_params(store_cigar=True, original_sequence_coordinates=orig, **kw)
        res = g.query_batch(ranges, pc)
        for fmt in ("paf", "bedpe"):
            want = "".join(c.query_paf(c.seq_name(t), s, e, range_name=rnames[i], merge_distance=20, fmt=fmt,
                                       original_sequence_coordinates=orig, **kw) for i, (t, s, e) in enumerate(ranges))
            assert res.paf(rnames, merge_distance=20, params=pc, fmt=fmt) == want
            if orig:
                assert "chrA\t" in want and "chrA:1000-21000\t" not in want.replace("an:Z:", "")
    # the CLI flag
    cli = os.path.join(os.path.dirname(impg_amd.__file__), "impg-gpu")
    paf = str(tmp_path / "t.paf")
    t, s, e = ranges[0]
    rng = "%s:%d-%d" % (c.seq_name(t), s, e)
    for fmt in ("bed", "paf"):
        r = subprocess.run([cli, "query", "-a", paf, "-r", rng, "-d", "20", "-x", "--min-transitive-len", "40", "-o", fmt,
                            "--original-sequence-coordinates"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        if fmt == "bed":
            want = c.query_bed(c.seq_name(t), s, e, merge_distance=20, original_sequence_coordinates=True, **kw)
        else:
            want = c.query_paf(c.seq_name(t), s, e, merge_distance=20, fmt="paf", original_sequence_coordinates=True, **kw)
        assert r.stdout == want


def test_parallel_result_assembly(tmp_path):
    """Large batches assemble their result lists range-parallel on the host (runs of one range per level); small
    chunks take the serial passes.  Same rows and CIGARs either way, and the oracle agrees on a sample."""
    text, names = random_paf(151, 3000, n_seq=4, seq_len=30000, self_aln=True, max_ops=40)
    g, c = both(tmp_path, text)
    ranges = random_ranges(17, 1500, 4, 30000, max_len=4000, min_len=120)
    for kw in (dict(transitive=True, max_depth=2, min_transitive_len=50), dict(transitive=True, dfs=True, max_depth=2),
               dict(), dict(transitive=True, max_depth=2, multi_impg=True)):
        p = impg_amd.make_params(store_cigar=True, **kw)
        g.set_option("chunk_ranges", 0)
        big = g.query_batch(ranges, p)
        assert len(big.intervals) > (1 << 18) or not kw.get("transitive")
        g.set_option("chunk_ranges", 40)
        small = g.query_batch(ranges, p)
        g.set_option("chunk_ranges", 0)
        assert big.offsets.tolist() == small.offsets.tolist()
        assert (big.intervals == small.intervals).all()
        for i in range(0, len(ranges), 97):
            t, s, e = ranges[i]
            want, wcg = c.query_cigar(t, s, e, **kw)
            assert big[i].tolist() == want.tolist()
            assert [x.tolist() for x in big.cigars(i)] == [x.tolist() for x in wcg]
            assert [x.tolist() for x in small.cigars(i)] == [x.tolist() for x in wcg]


@pytest.mark.parametrize("order", [impg_amd.ORDER_COITREES, impg_amd.ORDER_SORTED])
def test_adversarial_visit_order_equal_starts(tmp_path, order):
    """Targets with 9..64 (and a few larger) entries, most of them sharing their start: the visit order is then
    decided by the tie rule (input order) and by the order policy alone.  The transitive results depend on it
    through the order-dependent visited-set update (impg.rs:2471-2560); both policies have an exact checker
    (the oracle's coitrees restatement / its sorted-visits switch)."""
    rng = np.random.default_rng(99)
    lines, sizes = [], list(range(9, 65)) + [65, 100, 129, 500]
    for k, n in enumerate(sizes):
        tname = "T%d" % k
        starts = rng.choice([100, 100, 100, 400, 400, 900], size=n)
        for i in range(n):
            ts = int(starts[i])
            ln = int(rng.integers(150, 1200))
            qs = int(rng.integers(0, 50000))
            q = "T%d" % int(rng.integers(0, len(sizes)))
            if q == tname:
                q = "T%d" % ((k + 1) % len(sizes))
            lines.append("%s\t60000\t%d\t%d\t%s\t%s\t60000\t%d\t%d\t1\t1\t60\tcg:Z:%d=" %
                         (q, qs, qs + ln, "+-"[i % 2], tname, ts, ts + ln, ln))
    o.set_sorted_visits(order == impg_amd.ORDER_SORTED)
    try:
        g, c = both(tmp_path, "\n".join(lines) + "\n", order=order, bidirectional=False)
        ranges = []
        for k in range(len(sizes)):
            tid = g.seq_id("T%d" % k)
            ranges += [(tid, 0, 2000), (tid, 350, 450), (tid, 90, 110)]
        assert_same(g, c, ranges)
        assert_same(g, c, ranges, transitive=True, max_depth=3, min_transitive_len=20, min_distance_between_ranges=50)
        assert_same(g, c, ranges, transitive=True, dfs=True, max_depth=2, min_transitive_len=20)
        g2, c2 = both(tmp_path, "\n".join(lines) + "\n", order=order, bidirectional=True)
        assert_same(g2, c2, ranges[:60], transitive=True, max_depth=2, min_transitive_len=50, min_distance_between_ranges=200)
    finally:
        o.set_sorted_visits(False)


def test_cli_whole_sequence_strandness_and_stale_cache(tmp_path):
    """`-r name` without an interval is the whole sequence (main.rs:7290-7310); --consider-strandness keeps the
    strands apart in the BED merge (main.rs:4395-4409); a damaged -i cache is rebuilt from -a instead of
    failing every later run."""
    import os, subprocess
    text, names = random_paf(61, 200, n_seq=4, seq_len=20000)
    paf = str(tmp_path / "c.paf")
    open(paf, "w").write(text)
    cli = os.path.join(os.path.dirname(impg_amd.__file__), "impg-gpu")
    c = o.OracleIndex(paf_paths=[paf], preparse=True)
    for extra, kw in (([], dict()), (["--consider-strandness"], dict(consider_strandness=True)),
                      (["-x", "-m", "2", "--consider-strandness"], dict(transitive=True, max_depth=2, consider_strandness=True))):
        r = subprocess.run([cli, "query", "-a", paf, "-r", "s1", "-d", "50", "-o", "bed"] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert r.stdout == c.query_bed("s1", 0, 20000, "s1:0-20000", 50, o.make_params(**kw)), extra
    cache = str(tmp_path / "cache.idx")
    open(cache, "wb").write(b"IMPGHBM1" + bytes(100))  # a truncated / foreign cache file
    r = subprocess.run([cli, "query", "-a", paf, "-i", cache, "-r", "s2:100-5000", "-d", "0", "-o", "bed"], capture_output=True, text=True)
    assert r.returncode == 0 and "rebuilding" in r.stderr
    assert r.stdout == c.query_bed("s2", 100, 5000, "s2:100-5000", 0, o.make_params())
    r2 = subprocess.run([cli, "query", "-i", cache, "-r", "s2:100-5000", "-d", "0", "-o", "bed"], capture_output=True, text=True)
    assert r2.returncode == 0 and r2.stdout == r.stdout and "rebuilding" not in r2.stderr  # the rewritten cache loads
    assert not [f for f in os.listdir(tmp_path) if ".tmp." in f]


@pytest.mark.parametrize("seed,n_aln,mdbr", [(1, 900, 10), (2, 4000, 0), (3, 2500, 120)])
def test_visited_update_big_groups(tmp_path, seed, n_aln, mdbr):
    """(query, sequence) groups with hundreds to thousands of hits per level and visited lists of hundreds of
    ranges: the wave-per-group update (lists and pieces in LDS up to their caps, in global memory beyond), against
    the oracle's literal replay.  Few sequences, many short alignments, small min_transitive_len, depth 4."""
    rng = np.random.default_rng(seed)
    L = 400_000
    names = ["A", "B", "C"]
    lines = []
    for _ in range(n_aln):
        t, q = rng.choice(3, size=2, replace=False)
        ln = int(rng.integers(30, 400))
        ts, qs = int(rng.integers(0, L - ln)), int(rng.integers(0, L - ln))
        lines.append("%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t60\tcg:Z:%d=" %
                     (names[q], L, qs, qs + ln, "+-"[int(rng.integers(0, 2))], names[t], L, ts, ts + ln, ln, ln, ln))
    g, c = both(tmp_path, "\n".join(lines) + "\n")
    ranges = [(g.seq_id("A"), 0, L), (g.seq_id("B"), 1000, L - 1000), (g.seq_id("C"), 0, L // 2), (g.seq_id("A"), 50_000, 90_000)]
    kw = dict(transitive=True, max_depth=4, min_transitive_len=5, min_distance_between_ranges=mdbr)
    res = assert_same(g, c, ranges, **kw)
    assert len(res[0]) > n_aln  # every alignment is reached, most of them several times
    assert_same(g, c, ranges, transitive=True, dfs=True, max_depth=3, min_transitive_len=5, min_distance_between_ranges=mdbr)
    masked = {int(g.seq_id("B")): (L, [(k * 3000, k * 3000 + 900) for k in range(100)])}  # a mask list of 100 ranges: a big group from its first touch
    assert_same(g, c, ranges[:2], masked_regions=masked, **kw)
    # the pre-pass that drops hits the level's old lists already cover (option filter_covered: 0 never, 1 always)
    for f in (0, 1):
        g.set_option("filter_covered", f)
        assert_same(g, c, ranges, **kw)
        assert_same(g, c, ranges[:2], masked_regions=masked, **kw)


@pytest.mark.parametrize("fastga,seed", [(False, 1), (False, 2), (True, 3), (True, 4)])
def test_tracepoint_approximate_mode(fastga, seed):
    """Approximate mode on tracepoint alignments (scan_overlapping_tracepoints + project_overlapping_interval_fast,
    impg.rs:646-823, :1317-1533): the prefix-sum kernel against the oracle's literal segment scan -- Standard and
    FASTGA tracepoints, zero-length segments on either axis, both strands, forward and reversed entries,
    plain / transitive BFS / DFS / MultiImpg-flavoured queries, identity filter."""
    from tests.tp_gen import random_tp
    d = random_tp(seed, 700, n_seq=5, seq_len=60_000, fastga=fastga, self_aln=(seed % 2 == 0))
    g = impg_amd.GpuImpg.from_tracepoints(d["records"], d["tracepoints"], d["seq_len"], query_deltas=d["query_deltas"], diffs=d["diffs"],
                                          fastga=d["fastga"], trace_spacing=d["trace_spacing"], max_complexity=d["max_complexity"])
    c = o.OracleIndex(tracepoints=d)
    ranges = random_ranges(seed, 120, 5, 60_000, max_len=4000, min_len=1)
    assert_same(g, c, ranges)
    assert_same(g, c, ranges, min_identity=0.85)
    assert_same(g, c, ranges[:60], transitive=True, max_depth=3, min_transitive_len=30)
    assert_same(g, c, ranges[:60], transitive=True, max_depth=2, min_transitive_len=30, min_identity=0.8)
    assert_same(g, c, ranges[:40], transitive=True, dfs=True, max_depth=2, min_transitive_len=50)
    assert_same(g, c, ranges[:40], transitive=True, max_depth=2, multi_impg=True, min_transitive_len=50)
    with pytest.raises(impg_amd.ImpgGpuError) as ei:
        g.query_batch(ranges[:2], impg_amd.make_params(store_cigar=True))
    assert ei.value.code == impg_amd.IMPG_E_UNSUPPORTED
    # the same alignments sharded over three ranks (one handle): owners answer in approximate mode, hits come home
    gm = impg_amd.GpuImpg.from_tracepoints(d["records"], d["tracepoints"], d["seq_len"], query_deltas=d["query_deltas"], diffs=d["diffs"],
                                           fastga=d["fastga"], trace_spacing=d["trace_spacing"], max_complexity=d["max_complexity"],
                                           devices=[0, 0, 0], lanes=2)
    gm.set_option("chunk_ranges", 11)
    assert gm.approximate()
    assert_same(gm, c, ranges[:50])
    assert_same(gm, c, ranges[:50], transitive=True, max_depth=3, min_transitive_len=30)
    assert_same(gm, c, ranges[:30], transitive=True, dfs=True, max_depth=2, min_transitive_len=50, min_identity=0.8)
    del gm
    # the mode belongs to the index: the trait-shaped calls refuse the other one instead of answering in it
    assert g.approximate()
    t0, s0, e0 = ranges[0]
    assert g.query(t0, s0, e0, approximate_mode=True).tolist() == c.query(t0, s0, e0).tolist()
    assert g.query(t0, s0, e0).tolist() == c.query(t0, s0, e0).tolist()  # (default: the index's own mode, as the C entry points have it)
    for call in (lambda: g.query(t0, s0, e0, approximate_mode=False), lambda: g.query_transitive_bfs(t0, s0, e0, approximate_mode=False),
                 lambda: g.query_transitive_dfs(t0, s0, e0, approximate_mode=False)):
        with pytest.raises(impg_amd.ImpgGpuError) as ei:
            call()
        assert ei.value.code == impg_amd.IMPG_E_UNSUPPORTED
    # the saved index keeps its mode
    import tempfile, os
    with tempfile.TemporaryDirectory() as td:
        g.save(os.path.join(td, "tp.idx"))
        g2 = impg_amd.GpuImpg.load(os.path.join(td, "tp.idx"))
        assert_same(g2, c, ranges[:30], transitive=True, max_depth=2, min_transitive_len=30)
    # inputs outside the supported domain are refused, not mis-projected
    bad = d["tracepoints"].copy()
    bad[3] = -5
    with pytest.raises(impg_amd.ImpgGpuError):
        impg_amd.GpuImpg.from_tracepoints(d["records"], bad, d["seq_len"], query_deltas=d["query_deltas"], diffs=d["diffs"],
                                          fastga=d["fastga"], trace_spacing=d["trace_spacing"], max_complexity=d["max_complexity"])


@pytest.mark.parametrize("shuffle", [0, 12345])
def test_load_reference_impg_index(tmp_path, shuffle):
    """impg_gpu_index_load_impg: the reference's IMPGIDX2 file (written by the oracle's restatement of
    serialize_with_forest_map, impg.rs:1655-1721) + the PAF files it points into -> HBM index.  The engine and the
    oracle read the SAME file; with shuffle the file lists every tree's intervals in a scrambled order, as a real
    file does (coitrees' layout order), and both rebuild their tie order from it."""
    texts = [random_paf(70 + k, 150, n_seq=6, seq_len=50_000, self_aln=True)[0] for k in range(2)]
    pafs = []
    for k, t in enumerate(texts):
        p = str(tmp_path / ("f%d.paf" % k))
        open(p, "w").write(t)
        pafs.append(p)
    src = o.OracleIndex(paf_paths=pafs)
    f = str(tmp_path / "idx.impg")
    src.write_impg(f, shuffle_seed=shuffle)
    c = o.OracleIndex(impg_path=f, paf_paths=pafs)
    g = impg_amd.GpuImpg.load_impg(f, pafs)
    assert g.num_seqs() == c.num_seqs() and g.num_records() == 300
    for i in range(g.num_seqs()):
        assert g.seq_name(i) == c.seq_name(i) and g.seq_len(i) == c.seq_len(i)
    ranges = random_ranges(9, 80, 6, 50_000, max_len=6000, min_len=10)
    assert_same(g, c, ranges)
    assert_same(g, c, ranges, transitive=True, max_depth=3, min_transitive_len=20)
    assert_same(g, c, ranges[:40], transitive=True, dfs=True, max_depth=2)
    assert_same(g, c, ranges[:40], store_cigar=True, transitive=True, max_depth=2, min_transitive_len=40)
    assert not g.approximate()
    with pytest.raises(impg_amd.ImpgGpuError) as ei:  # (a CIGAR index has no approximate mode to offer)
        g.query(*ranges[0], approximate_mode=True)
    assert ei.value.code == impg_amd.IMPG_E_UNSUPPORTED
    # the CLI opens the same file with -i ... -a ...
    import os, subprocess
    cli = os.path.join(os.path.dirname(impg_amd.__file__), "impg-gpu")
    r = subprocess.run([cli, "query", "-i", f, "-a"] + pafs + ["-r", "s1:1000-9000", "-d", "100", "-x", "-m", "2", "-o", "bed"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == c.query_bed("s1", 1000, 9000, "s1:1000-9000", 100, o.make_params(transitive=True, max_depth=2))
    assert open(f, "rb").read()[:8] == b"IMPGIDX2"  # (never overwritten by the library's own cache format)
    # a PAF that changed under the index is noticed
    open(pafs[0], "w").write("x" * 100 + "\n" + texts[0])
    with pytest.raises(impg_amd.ImpgGpuError):
        impg_amd.GpuImpg.load_impg(f, pafs)
    with pytest.raises(impg_amd.ImpgGpuError):
        impg_amd.GpuImpg.load_impg(pafs[1], pafs)


def _bed_three_ways(g, c, ranges, names, d, **kw):
    """device-side merge == host-side merge == the oracle's perform_query + output_results_bed, byte for byte"""
    p = impg_amd.make_params(**kw)
    dev = g.query_batch_bed(ranges, p, merge_distance=d, range_names=names)
    host = g.query_batch(ranges, p).bed(names, merge_distance=d, params=p)
    assert dev == host, (d, kw)
    okw = dict(kw)
    want = "".join(c.query_bed(c.seq_name(t), s, e, range_name=(names[i] if names else None), merge_distance=d, **okw)
                   for i, (t, s, e) in enumerate(ranges))
    assert dev == want, (d, kw)
    return dev


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_device_bed_merge_random(tmp_path, seed):
    text, _ = random_paf(seed, 500, n_seq=5, seq_len=30000, self_aln=True, weird=(seed == 13))
    g, c = both(tmp_path, text)
    ranges = random_ranges(seed, 60, 5, 30000, max_len=5000, min_len=150)
    names = [None if i % 3 else "n%d" % i for i in range(len(ranges))]
    names_o = [("n%d" % i) if i % 3 == 0 else None for i in range(len(ranges))]
    assert names == names_o
    for d in (-1, 0, 25, 1000, 100000):
        for kw in (dict(), dict(transitive=True, max_depth=3, min_transitive_len=20), dict(transitive=True, dfs=True, max_depth=2),
                   dict(transitive=True, max_depth=2, multi_impg=True), dict(min_output_length=400),
                   dict(transitive=True, max_depth=2, min_output_length=300), dict(transitive=True, max_depth=3, consider_strandness=True),
                   dict(consider_strandness=True)):
            _bed_three_ways(g, c, ranges, names, d, **kw)
    # chunking does not show
    g.set_option("chunk_ranges", 7)
    g.set_option("pair_budget", 2048)
    _bed_three_ways(g, c, ranges, names, 50, transitive=True, max_depth=3, min_transitive_len=20)


def test_device_bed_merge_chains_and_worst_case(tmp_path):
    """gap_2d chains: collinear alignments on one (query, target, strand) at gaps just below / above d on either axis,
    on both strands, interleaved with off-diagonal ones -- and the O(k^2) worst case: thousands of overlapping hits
    in one group, every row within d of every other, so no early break ever fires."""
    lines = []
    L = 2_000_000
    pos_q, pos_t = 1000, 5000
    rng = np.random.default_rng(3)
    for k in range(300):  # a forward chain with gaps around d = 50 on either axis
        ln = int(rng.integers(80, 200))
        lines.append("Q\t%d\t%d\t%d\t+\tT\t%d\t%d\t%d\t%d\t%d\t60\tcg:Z:%d=" % (L, pos_q, pos_q + ln, L, pos_t, pos_t + ln, ln, ln, ln))
        pos_q += ln + int(rng.choice([0, 10, 49, 50, 51, 200]))
        pos_t += ln + int(rng.choice([0, 10, 49, 50, 51, 200]))
    pos_q, pos_t = 900_000, 400_000
    for k in range(300):  # a reverse-strand chain: the query runs backwards along the target
        ln = int(rng.integers(80, 200))
        lines.append("Q\t%d\t%d\t%d\t-\tT\t%d\t%d\t%d\t%d\t%d\t60\tcg:Z:%d=" % (L, pos_q - ln, pos_q, L, pos_t, pos_t + ln, ln, ln, ln))
        pos_q -= ln + int(rng.choice([0, 10, 49, 50, 51, 200]))
        pos_t += ln + int(rng.choice([0, 10, 49, 50, 51, 200]))
    for k in range(3000):  # the worst case: one (Q2, T) group, all hits inside one 3 kb window of both axes
        ln = int(rng.integers(500, 2500))
        a = int(rng.integers(0, 3000 - 400))
        lines.append("Q2\t%d\t%d\t%d\t%s\tT\t%d\t%d\t%d\t%d\t%d\t60\tcg:Z:%d=" %
                     (L, 10_000 + a, 10_000 + a + ln, "+-"[k % 2], L, 1_000_000 + a, 1_000_000 + a + ln, ln, ln, ln))
    g, c = both(tmp_path, "\n".join(lines) + "\n", bidirectional=False)
    T = g.seq_id("T")
    ranges = [(T, 0, 700_000), (T, 1_000_000, 1_003_000), (T, 0, L), (T, 5000, 5200)]
    for d in (0, 49, 50, 51, 5000):
        _bed_three_ways(g, c, ranges, None, d)
        _bed_three_ways(g, c, ranges, None, d, consider_strandness=True)


def _device_rows_by_range(dr, n_ranges, min_output_length=None):
    """The slots impg_gpu_query_batch_device left in HBM, attributed through source[] / frontier[]: per range of the batch
    the list of (query_id, q_first, q_last, target_id, t_first, t_last) rows, in no particular order."""
    rows = [[] for _ in range(n_ranges)]
    for k in range(len(dr.parts())):
        first, level, qid, co, src, fr = dr.part_to_host(k)
        live = qid != np.uint32(0xFFFFFFFF)
        if min_output_length is not None:
            live &= np.abs(co[:, 1].astype(np.int64) - co[:, 0]) >= min_output_length
        assert (src < len(fr)).all()
        f = fr[src[live]]
        for q, r, tg in zip((first + f["range_idx"]).tolist(), np.column_stack([qid[live], co[live]]).tolist(), f["target_id"].tolist()):
            rows[q].append((r[0], r[1], r[2], tg, r[3], r[4]))
    return rows


def test_device_rows_attributed(tmp_path):
    """impg_gpu_query_batch_device: every row left in HBM belongs to a range of the batch and carries its target -- the
    multiset of a range's rows equals the oracle's (impg.rs:2491-2503 pushes exactly these), whichever kernels lay the final
    level out (fused entry by entry or listed), and the counts / checksums recomputed from those rows equal the counting
    form's.  The emission ORDER is not part of this layout (IMPG_ROWS_ATTRIBUTED)."""
    from tests.test_gpu_fullsize import checksum
    for seed, kwp in [(311, dict(n_seq=6, seq_len=20000, weird=True, self_aln=True)), (312, dict(n_seq=60, seq_len=4000, self_aln=True))]:
        text, names = random_paf(seed, 900, **kwp)
        g, c = both(tmp_path, text)
        ranges = random_ranges(seed + 1, 120, kwp["n_seq"], kwp["seq_len"], max_len=kwp["seq_len"] // 5, min_len=40)
        kws = [dict(), dict(transitive=True, max_depth=1, min_transitive_len=20),
               dict(transitive=True, max_depth=3, min_transitive_len=20, min_distance_between_ranges=0),
               dict(transitive=True, max_depth=0, min_transitive_len=30, min_output_length=60),
               dict(transitive=True, max_depth=3, min_identity=0.7)]
        for kw in (kws if seed == 311 else kws[:1] + kws[2:4]):
            p = impg_amd.make_params(**kw)
            want, n_proj = [], 0
            for (t, s, e) in ranges:
                want.append(c.query(t, s, e, **kw))
                n_proj += c.last_projection_count()  # (every Some(..), rows below min_output_length included: impg.rs:2482-2504)
            for lm, fuse, chunk in [(1, 1, 0), (1, 0, 0), (4096, 1, 0), (1, 1, 37)]:
                g.set_option("locality_min", lm)
                g.set_option("fuse_final_level", fuse)
                g.set_option("chunk_ranges", chunk)
                dr = g.query_batch_device(ranges, p)
                st, cnt, ck = g.query_batch_stats(ranges, p)
                assert dr.projected == st.projected == n_proj
                cnt2, ck2 = dr.check()
                assert (cnt2 == cnt).all() and (ck2 == ck).all(), (seed, kw, lm, fuse, chunk)
                got = _device_rows_by_range(dr, len(ranges), kw.get("min_output_length") if kw.get("transitive") else None)
                for i in range(len(ranges)):
                    assert sorted(got[i]) == sorted(tuple(int(x) for x in r) for r in want[i][1:].tolist()), (seed, kw, lm, fuse, chunk, i)
                    assert int(ck2[i]) == checksum(want[i][1:])
                dr.free()
            g.set_option("chunk_ranges", 0)
            g.set_option("locality_min", 4096)
            g.set_option("fuse_final_level", 1)
            # IMPG_ROWS_ORDERED: the trait's rows themselves, left in HBM -- the oracle's rows in the oracle's order
            for lm, chunk in [(1, 0), (4096, 0), (1, 37)]:
                g.set_option("locality_min", lm)
                g.set_option("chunk_ranges", chunk)
                do = g.query_batch_device(ranges, p, layout=impg_amd._lib.ROWS_ORDERED)
                assert do.projected == n_proj
                seen = 0
                for k in range(len(do.parts())):
                    first, rows, off = do.ordered_to_host(k)
                    for j in range(len(off) - 1):
                        assert rows[off[j]:off[j + 1]].tolist() == want[first + j].tolist(), (seed, kw, lm, chunk, first + j)
                    seen += len(off) - 1
                assert seen == len(ranges)
                do.free()
                # IMPG_ROWS_ORDERED_SLOTS: every slot at its place; with the hole rows taken out, the same rows in the same order
                ds = g.query_batch_device(ranges, p, layout=impg_amd._lib.ROWS_ORDERED_SLOTS)
                assert ds.projected == n_proj
                seen = 0
                for k in range(len(ds.parts())):
                    first, rows, off = ds.ordered_to_host(k)
                    for j in range(len(off) - 1):
                        r = rows[off[j]:off[j + 1]]
                        r = r[r["query_id"] != 0xFFFFFFFF]
                        assert r.tolist() == want[first + j].tolist(), (seed, kw, lm, chunk, first + j)
                    seen += len(off) - 1
                assert seen == len(ranges)
                ds.free()
            g.set_option("chunk_ranges", 0)
            g.set_option("locality_min", 4096)
        # what the layout does not take is refused, not answered some other way
        for kw in [dict(transitive=True, dfs=True), dict(store_cigar=True), dict(transitive=True, multi_impg=True)]:
            with pytest.raises(impg_amd.ImpgGpuError) as e:
                g.query_batch_device(ranges[:4], impg_amd.make_params(**kw))
            assert e.value.code == impg_amd.IMPG_E_UNSUPPORTED


@pytest.mark.parametrize("order", [impg_amd.ORDER_COITREES, impg_amd.ORDER_SORTED])
def test_windows_of_thousands_of_hits(tmp_path, order):
    """A repeat hot spot: 9 500 alignments piled on one stretch of a target, so that a range's window holds more hits than the
    block-per-window emit's 4 096-key buffer (lookup_emit_wide_kernel: rank bins gathered into groups, a collect-sort-write pass
    per group) -- and ranges beside it with a few hundred and a few dozen.  Rows in visit order against the oracle, plain and
    transitive, listed and fused final levels, the rows left in HBM in both ordered layouts."""
    rng = np.random.default_rng(77)
    L = 400_000
    lines = []
    for i in range(9500):
        a = int(rng.integers(100_000, 100_400))
        ln = int(rng.integers(300, 900))
        q = "Q%d" % (i % 37)
        qa = int(rng.integers(0, L - 2000))
        lines.append("%s\t%d\t%d\t%d\t%s\tT\t%d\t%d\t%d\t%d\t%d\t60\tcg:Z:%d=" % (q, L, qa, qa + ln, "+-"[i % 2], L, a, a + ln, ln, ln, ln))
    for i in range(600):  # a milder pile and a sparse stretch on the same target
        a = int(rng.integers(200_000, 203_000)) if i < 450 else int(rng.integers(250_000, 390_000))
        ln = int(rng.integers(200, 700))
        qa = int(rng.integers(0, L - 2000))
        lines.append("Q%d\t%d\t%d\t%d\t+\tT\t%d\t%d\t%d\t%d\t%d\t60\tcg:Z:%d=" % (i % 37, L, qa, qa + ln, L, a, a + ln, ln, ln, ln))
    g, c = both(tmp_path, "\n".join(lines) + "\n", order=order, bidirectional=False)
    if order == impg_amd.ORDER_SORTED:
        o.set_sorted_visits(True)
    try:
        T = g.seq_id("T")
        ranges = [(T, 100_350, 100_420), (T, 100_000, 101_500), (T, 200_500, 201_000), (T, 300_000, 300_900), (T, 100_399, 100_401)] * 3
        for kw in [dict(), dict(transitive=True, max_depth=1, min_transitive_len=10), dict(transitive=True, max_depth=2, min_transitive_len=10)]:
            p = impg_amd.make_params(**kw)
            want = [c.query(t, s, e, **kw) for (t, s, e) in ranges]
            assert max(len(w) for w in want) > 4096 + 1
            for lm in (1, 4096):
                g.set_option("locality_min", lm)
                res = g.query_batch(ranges, p)
                for i in range(len(ranges)):
                    assert res[i].tolist() == want[i].tolist(), (kw, lm, i)
                st, cnt, ck = g.query_batch_stats(ranges, p)
                assert cnt.tolist() == [len(w) - 1 for w in want]
                for layout in (impg_amd._lib.ROWS_ORDERED, impg_amd._lib.ROWS_ORDERED_SLOTS):
                    d = g.query_batch_device(ranges, p, layout=layout)
                    first, rows, off = d.ordered_to_host(0)
                    d.free()
                    for i in range(len(ranges)):
                        r = rows[off[i]:off[i + 1]]
                        assert r[r["query_id"] != 0xFFFFFFFF].tolist() == want[i].tolist(), (kw, lm, layout, i)
            g.set_option("locality_min", 4096)
    finally:
        o.set_sorted_visits(False)


def test_counting_runs_with_slots_in_projection_order(tmp_path):
    """A counting run (nothing kept) under the lookup order lays its hit slots out in projection order, not the
    reference's: per-range counts and checksums must equal those of the full results -- which keep the
    reference's slot order -- at every depth, BFS and DFS, whichever way the option is set."""
    from tests.test_gpu_fullsize import checksum
    for seed, kwp in [(211, dict(n_seq=6, seq_len=20000, weird=True, self_aln=True)), (212, dict(n_seq=60, seq_len=4000, self_aln=True))]:
        text, names = random_paf(seed, 900, **kwp)
        g, c = both(tmp_path, text)
        ranges = random_ranges(seed + 1, 200, kwp["n_seq"], kwp["seq_len"], max_len=kwp["seq_len"] // 5, min_len=40)
        for kw in [dict(), dict(transitive=True, max_depth=4, min_transitive_len=20, min_distance_between_ranges=0),
                   dict(transitive=True, max_depth=0, min_transitive_len=30),
                   dict(transitive=True, dfs=True, max_depth=3, min_transitive_len=20),
                   dict(transitive=True, max_depth=3, min_identity=0.7)]:
            p = impg_amd.make_params(**kw)
            g.set_option("locality_min", 4096)
            res = g.query_batch(ranges, p)
            want_cnt = [len(res[i]) - 1 for i in range(len(ranges))]
            want_ck = [checksum(res[i][1:]) for i in range(len(ranges))]
            for lm, free, regroup, fuse in [(1, 1, 1, 1), (1, 1, 1, 0), (1, 0, 1, 1), (0, 1, 1, 1), (1, 1, 0, 1), (0, 0, 0, 0)]:
                g.set_option("locality_min", lm)
                g.set_option("free_slot_order", free)
                g.set_option("regroup_entries", regroup)  # (a projection block sorts its pairs by entry first: same slots, same rows)
                g.set_option("fuse_final_level", fuse)    # (the final level's pairs enumerated from the count pass's windows: no emit pass)
                st, cnt, ck = g.query_batch_stats(ranges, p)
                assert st.projected == res.projected
                assert cnt.tolist() == want_cnt, (seed, kw, lm, free, regroup, fuse)
                assert [int(x) for x in ck] == want_ck, (seed, kw, lm, free, regroup, fuse)
                st2, _, _ = g.query_batch_stats(ranges, p, counts=False, checksums=False)  # (no per-range list is written at all)
                assert st2.projected == res.projected
            g.set_option("fuse_final_level", 1)
            g.set_option("regroup_entries", 0)
            g.set_option("locality_min", 4096)
            res0 = g.query_batch(ranges, p)
            g.set_option("regroup_entries", 1)
            assert res0.projected == res.projected
            for i in range(len(ranges)):
                assert res0[i].tolist() == res[i].tolist(), (seed, kw, i)
        g.set_option("free_slot_order", 1)
    # dense target: windows wider than 64 entries are listed by place for the wave-per-range emit pass
    lines = []
    for i in range(300):
        lines.append("Q%d\t9000\t%d\t%d\t%s\tT\t9000\t%d\t%d\t10\t10\t60\tcg:Z:%d=" %
                     (i % 11, 100 + i, 1100 + i, "+-"[i % 2], 2000 + (i * 7) % 900, 3000 + (i * 7) % 900, 1000))
    g, c = both(tmp_path, "\n".join(lines) + "\n")
    t = g.seq_id("T")
    dense = [(t, 2500, 2600), (t, 2000, 4000), (t, 2890, 2910), (g.seq_id("Q3"), 0, 2000)] * 20
    for kw in [dict(), dict(transitive=True, max_depth=2, min_transitive_len=10)]:
        p = impg_amd.make_params(**kw)
        res = g.query_batch(dense, p)
        for lm, free, fuse in [(1, 1, 1), (1, 1, 0), (1, 0, 1)]:
            g.set_option("locality_min", lm)
            g.set_option("free_slot_order", free)
            g.set_option("fuse_final_level", fuse)
            st, cnt, ck = g.query_batch_stats(dense, p)
            assert st.projected == res.projected
            assert cnt.tolist() == [len(res[i]) - 1 for i in range(len(dense))]
            assert [int(x) for x in ck] == [checksum(res[i][1:]) for i in range(len(dense))]
    # runs of ranges without a single hit between the ones with hits (the final level's tiles then span more than the 64
    # ranges a projection block looks at in one go), unknown targets, and windows wider than the hit mask in the same batch
    lines = lines[:120] + ["E%d\t5000\t0\t100\t+\tE%d\t5000\t4000\t4100\t10\t10\t60\tcg:Z:100=" % (i, i + 1) for i in range(0, 40, 2)]
    g, c = both(tmp_path, "\n".join(lines) + "\n")
    t = g.seq_id("T")
    gaps = []
    for i in range(12):
        gaps += [(g.seq_id("E%d" % (2 * (j % 20))), 1000 + j, 1500 + j) for j in range(90 + 17 * i)]   # no alignment there
        gaps += [(t, 2000 + 31 * i, 2300 + 31 * i), (g.seq_id("Q%d" % (i % 11)), 0, 3000), (g.num_seqs() + 5, 10, 20)]
    g.set_option("locality_min", 1)
    g.set_option("free_slot_order", 1)
    for kw in [dict(), dict(transitive=True, max_depth=2, min_transitive_len=10), dict(transitive=True, max_depth=1)]:
        p = impg_amd.make_params(**kw)
        res = g.query_batch(gaps, p)
        for fuse in (1, 0):
            g.set_option("fuse_final_level", fuse)
            st, cnt, ck = g.query_batch_stats(gaps, p)
            assert st.projected == res.projected
            assert cnt.tolist() == [len(res[i]) - 1 for i in range(len(gaps))]
            assert [int(x) for x in ck] == [checksum(res[i][1:]) for i in range(len(gaps))]


def _cigar_spans(cg):
    """(target span, query span, target offsets of every op boundary) of a CIGAR string."""
    import re
    t = q = 0
    bounds = [0]
    for n, op in re.findall(r"(\d+)([=XIDM])", cg):
        n = int(n)
        if op != "I": t += n
        if op != "D": q += n
        bounds.append(t)
    return t, q, bounds


def test_prefix_line_edges(tmp_path):
    """The plain projection never replays ops: it locates the first / last overlapping op on the tiles' 16-bit
    prefix lines and verifies the candidates.  Ranges that start and end on, one before and
    one after EVERY op boundary of multi-tile records -- insertions and deletions sitting exactly on the range ends,
    candidates across tile borders (26 ops per tile) and across the thirds of a tile, both strands, both entry
    directions -- must project exactly as the oracle's op-by-op walk does; so must a record whose tiles overflow
    16 bits (literal walk) and one whose CIGAR disagrees with its PAF coordinates."""
    unit = "7=2I5=3D1X4=1I6=2D"                      # 9 ops; insertions and deletions between matches
    cg_a = unit * 9                                   # 81 ops: 4 tiles, borders inside the unit
    cg_b = ("11=1X" * 13 + "5I" + "9=4D" * 14)        # 55 ops: a tile border between two matches, an insertion at op 26
    cg_w = "30000=5I20000=7D25000=" + "3=1X" * 20     # the first tile sums to 75 000 on both axes: wide
    ta, qa, ba = _cigar_spans(cg_a)
    tb, qb, bb = _cigar_spans(cg_b)
    tw, qw, _ = _cigar_spans(cg_w)
    L = 200000
    lines = []
    def rec(q, qs, qspan, strand, t, ts, tspan, cg):
        lines.append("%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t1\t1\t60\tcg:Z:%s" % (q, L, qs, qs + qspan, strand, t, L, ts, ts + tspan, cg))
    rec("Q1", 500, qa, "+", "T", 1000, ta, cg_a)
    rec("Q2", 700, qa, "-", "T", 1000 + 3, ta, cg_a)
    rec("Q3", 100, qb, "+", "T", 1200, tb, cg_b)
    rec("Q4", 900, qb, "-", "T", 1100, tb, cg_b)
    rec("Q5", 2000, qw, "+", "T", 3000, tw, cg_w)
    rec("Q6", 2000, qw, "-", "T", 3500, tw, cg_w)
    rec("Q7", 50, qa, "+", "T", 900, ta + 40, cg_a)  # CIGAR shorter than the PAF target span: no end shortcut, odd tails
    g, c = both(tmp_path, "\n".join(lines) + "\n")
    T = g.seq_id("T")
    pts = sorted({1000 + b + d for b in ba for d in (-1, 0, 1)} | {1200 + b + d for b in bb for d in (-1, 0, 1)} |
                 {1003 + b for b in ba} | {1100 + b for b in bb})
    pts = [p for p in pts if p > 0]
    ranges = []
    for i, p in enumerate(pts):
        ranges.append((T, p, p + 1 + (i * 7) % 23))           # starts on / around a boundary
        ranges.append((T, max(0, p - 1 - (i * 5) % 31), p))   # ends on / around a boundary
    ranges += [(T, pts[i], pts[j]) for i in range(0, len(pts) - 40, 17) for j in (i + 9, i + 40)]
    ranges += [(T, 2990, 3010), (T, 32990, 33020), (T, 3000 + 30000, 3000 + 30001), (T, 52000, 54000), (T, 3000, 3000 + tw), (T, 78000, 78100)]
    # from the query side the reversed entries walk the same records (the reverse-strand ones back to front)
    for q, qs, qspan in [("Q1", 500, qa), ("Q2", 700, qa), ("Q3", 100, qb), ("Q4", 900, qb)]:
        qid = g.seq_id(q)
        for k in range(0, qspan, 3):
            ranges.append((qid, qs + k, qs + k + 1 + k % 19))
    for kw in [dict(), dict(transitive=True, max_depth=2, min_transitive_len=1, min_distance_between_ranges=0)]:
        for lo in range(0, len(ranges), 400):
            assert_same(g, c, ranges[lo:lo + 400], **kw)


def test_identity_filter_wide_last_tile(tmp_path):
    """min_gap_compressed_identity when an end of the slice is answered by the no-read shortcut (the range covers the
    alignment's start / end) and the record's LAST storage tile is `wide` (its sums pass 2^16, e.g. a 70000= op): the
    identity line's 16-bit fields do not hold that tile's sums, so the pair must take the literal walk (round-3 advisory:
    the flag was only tested inside the search, which a shortcut end skips).  Both strands, both entry directions,
    ranges covering the end, the start, the whole alignment and neither."""
    cgs = ["70000=10I70000=", "65536=5X", "3=1X" * 30 + "70000=10I70000=", "70000=10I70000=" + "3=1X" * 30, "40000=30000X5D"]
    L = 400000
    lines = []
    spans = []
    for i, cg in enumerate(cgs):
        t, q, _ = _cigar_spans(cg)
        for strand in "+-":
            ts = 1000 + 37 * i
            qs = 2000 + 11 * i
            lines.append("Q%d%s\t%d\t%d\t%d\t%s\tT\t%d\t%d\t%d\t1\t1\t60\tcg:Z:%s" % (i, "f" if strand == "+" else "r", L, qs, qs + q, strand, L, ts, ts + t, cg))
            spans.append(("Q%d%s" % (i, "f" if strand == "+" else "r"), qs, q, ts, t))
    g, c = both(tmp_path, "\n".join(lines) + "\n")
    T = g.seq_id("T")
    ranges = []
    for name, qs, q, ts, t in spans:
        qid = g.seq_id(name)
        for (a, b) in [(0, t + 5000), (ts, ts + t), (ts + t - 100, ts + t + 50), (ts + t - 70001, ts + t), (ts - 10, ts + 50), (ts + 10, ts + t - 10),
                       (ts + 69990, ts + 70020), (ts + 100, ts + t)]:
            ranges.append((T, max(0, a), min(L, b)))
        for (a, b) in [(0, qs + q + 100), (qs, qs + q), (qs + q - 100, qs + q + 10), (qs - 5, qs + 70005), (qs + 50, qs + q - 50), (qs + 69995, qs + q)]:
            ranges.append((qid, max(0, a), min(L, b)))
    ranges = [r for r in ranges if r[1] < r[2]]
    for thr in (0.5, 0.9999, 0.99995, 0.999929, 0.57, 1.0):
        assert_same(g, c, ranges, min_identity=thr)
    assert_same(g, c, ranges[:40], transitive=True, max_depth=2, min_transitive_len=1, min_identity=0.9999)


@pytest.mark.parametrize("seed,max_ops,bidirectional,order", [(1, 60, True, 0), (2, 700, True, 0), (3, 30, False, 1), (4, 2500, True, 0)])
def test_device_build_matches_host_build(tmp_path, seed, max_ops, bidirectional, order):
    """The index built by kernels from the packed ops (index_build_device.hip: op lines, prefix lines, identity
    prefixes, checkpoints inline and external, entries by (target, start), columns, levels) against the host builder
    (index_build.cpp, IMPG_BUILD_HOST=1): the two saved index files are the same bytes.  CIGARs of 1 .. 2 500 ops
    (external checkpoints from 209 on), zero-length and inconsistent ops, tiles whose sums overflow 16 bits, several
    alignment files (MultiImpg tie ranks), both order policies."""
    import os
    sl = 3_000_000 if max_ops > 1000 else 200_000
    texts = [random_paf(seed * 10 + k, 220, n_seq=6, seq_len=sl, max_ops=max_ops, weird=True, inconsistent=(seed % 2 == 0), self_aln=True)[0]
             for k in range(2)]
    # tiles whose running sums do not fit 16 bits: ops of 70 kb, and 40 ops of 3 kb in one tile
    texts[0] += "s1\t%d\t0\t140010\t+\ts2\t%d\t100\t140110\t140000\t140010\t60\tcg:Z:70000=10I70000=\n" % (sl, sl)
    texts[0] += "s0\t%d\t5\t120005\t-\ts3\t%d\t7\t120007\t120000\t120000\t60\tcg:Z:%s\n" % (sl, sl, "3000=" * 40)
    paths = []
    for k, t in enumerate(texts):
        paths.append(str(tmp_path / ("f%d.paf" % k)))
        open(paths[-1], "w").write(t)
    files = {}
    for mode in ("device", "host"):
        if mode == "host":
            os.environ["IMPG_BUILD_HOST"] = "1"
        try:
            g = impg_amd.GpuImpg.from_paf(paths, bidirectional=bidirectional, order=order)
        finally:
            os.environ.pop("IMPG_BUILD_HOST", None)
        f = str(tmp_path / (mode + ".idx"))
        g.save(f)
        files[mode] = open(f, "rb").read()
        del g
    assert len(files["device"]) == len(files["host"])
    if files["device"] != files["host"]:
        a, b = np.frombuffer(files["device"], dtype=np.uint8), np.frombuffer(files["host"], dtype=np.uint8)
        bad = np.nonzero(a != b)[0]
        raise AssertionError("saved indexes differ in %d bytes, first at offset %d of %d" % (len(bad), int(bad[0]), len(a)))


def test_index_without_prefix_lines(tmp_path):
    """An index that does not fit the device with its prefix lines is built without them (here forced with
    IMPG_PREFIX_LINES=0): the plain projection then takes the two short walks on the op lines.  Same rows as the oracle,
    40 % fewer bytes, device build == host build, and the saved file keeps the choice."""
    import os
    text, _ = random_paf(11, 400, n_seq=6, seq_len=60_000, max_ops=400, weird=True, inconsistent=True, self_aln=True)
    g_full, c = both(tmp_path, text)
    os.environ["IMPG_PREFIX_LINES"] = "0"
    try:
        g = impg_amd.GpuImpg.from_paf(str(tmp_path / "t.paf"))
        os.environ["IMPG_BUILD_HOST"] = "1"
        g_host = impg_amd.GpuImpg.from_paf(str(tmp_path / "t.paf"))
    finally:
        os.environ.pop("IMPG_PREFIX_LINES", None)
        os.environ.pop("IMPG_BUILD_HOST", None)
    assert g.device_bytes() < 0.85 * g_full.device_bytes()  # (op lines + sub-tile rows against op + prefix lines; the identity lines come on demand)
    ranges = random_ranges(5, 150, 6, 60_000, max_len=6000, min_len=1)
    assert_same(g, c, ranges)
    assert_same(g, c, ranges[:60], transitive=True, max_depth=3, min_transitive_len=20)
    assert_same(g, c, ranges[:40], transitive=True, dfs=True, max_depth=2, min_transitive_len=40)
    assert_same(g, c, ranges, min_identity=0.9)
    assert_same(g, c, ranges[:40], store_cigar=True, transitive=True, max_depth=2, min_transitive_len=40)
    assert g.query(*ranges[0]).tolist() == c.query(*ranges[0]).tolist()  # (the one-sync small-batch path)
    g.save(str(tmp_path / "np.idx"))
    g_host.save(str(tmp_path / "np_host.idx"))
    assert open(str(tmp_path / "np.idx"), "rb").read() == open(str(tmp_path / "np_host.idx"), "rb").read()
    g2 = impg_amd.GpuImpg.load(str(tmp_path / "np.idx"))
    assert g2.device_bytes() == g.device_bytes()
    assert_same(g2, c, ranges[:50], transitive=True, max_depth=2)


def test_identity_lines_on_demand(tmp_path):
    """An index with prefix lines carries no identity lines until a query filters by min_gap_compressed_identity
    (impg.rs:1283-1287): 2.2 KB a record instead of 3.2.  The lines built on demand, from the op lines, are byte for byte
    the ones both builders write when asked up front (IMPG_IDENTITY_LINES=1), through every entry point that can be the
    first to ask -- the batch engine, the one-synchronisation small batch, the per-query walk -- and a saved index keeps
    whichever state it was saved in."""
    import os
    text, _ = random_paf(31, 300, n_seq=6, seq_len=50_000, max_ops=500, weird=True, inconsistent=True, self_aln=True)
    g, c = both(tmp_path, text)
    paf = str(tmp_path / "t.paf")
    ranges = random_ranges(9, 120, 6, 50_000, max_len=5000, min_len=1)
    lean = g.device_bytes()
    assert_same(g, c, ranges, transitive=True, max_depth=2, min_transitive_len=20)  # (no filter: nothing is built)
    assert g.device_bytes() == lean
    g.save(str(tmp_path / "lean.idx"))
    assert_same(g, c, ranges, min_identity=0.9)  # the batch engine asks first
    full = g.device_bytes()
    assert full > lean and (full - lean) % 128 == 0  # one 128-byte line a tile
    assert_same(g, c, ranges[:40], transitive=True, max_depth=3, min_transitive_len=20, min_identity=0.85)
    g.save(str(tmp_path / "lazy.idx"))
    os.environ["IMPG_IDENTITY_LINES"] = "1"
    try:
        eager = impg_amd.GpuImpg.from_paf(paf)
        os.environ["IMPG_BUILD_HOST"] = "1"
        eager_host = impg_amd.GpuImpg.from_paf(paf)
    finally:
        os.environ.pop("IMPG_IDENTITY_LINES", None)
        os.environ.pop("IMPG_BUILD_HOST", None)
    assert eager.device_bytes() == full == eager_host.device_bytes()
    eager.save(str(tmp_path / "eager.idx"))
    eager_host.save(str(tmp_path / "eager_host.idx"))
    lazy_bytes = open(str(tmp_path / "lazy.idx"), "rb").read()
    assert lazy_bytes == open(str(tmp_path / "eager.idx"), "rb").read() == open(str(tmp_path / "eager_host.idx"), "rb").read()
    # a lean file loads lean and builds its lines like the index it was saved from; other first askers: the small batch, the walk
    h = impg_amd.GpuImpg.load(str(tmp_path / "lean.idx"))
    assert h.device_bytes() == lean
    t, s0, e0 = ranges[3]
    assert h.query(t, s0, e0, min_gap_compressed_identity=0.9).tolist() == c.query(t, s0, e0, min_identity=0.9).tolist()
    assert h.device_bytes() == full
    h2 = impg_amd.GpuImpg.load(str(tmp_path / "lean.idx"))
    got = h2.query_transitive_dfs(t, s0, e0, max_depth=2, min_transitive_len=20, min_gap_compressed_identity=0.9)
    assert got.tolist() == c.query(t, s0, e0, transitive=True, dfs=True, max_depth=2, min_transitive_len=20, min_identity=0.9).tolist()
    assert h2.device_bytes() == full
    assert impg_amd.GpuImpg.load(str(tmp_path / "lazy.idx")).device_bytes() == full


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_walk_kernel_matches_oracle_and_batch_engine(tmp_path, seed):
    """The per-query walk (walk_device.inc: a workgroup takes a query through all its pops / levels in one launch) against
    the oracle and against the batch engine: DFS batches of any size (the default), small BFS batches (walk_kernel = 2),
    the batch engine alone (0) -- rows, per-range counts and checksums; depth limits incl. unlimited, the distance and
    length cut-offs, the identity filter, the subset filter, ranges on sequences without alignments, dense targets."""
    text, _ = random_paf(40 + seed, 500, n_seq=7, seq_len=40_000, max_ops=300, weird=True, inconsistent=(seed == 2), self_aln=True)
    g, c = both(tmp_path, text)
    ranges = random_ranges(seed, 150, 7, 40_000, max_len=5000, min_len=1)
    keep = np.array([1, 0, 1, 1, 0, 1, 1], dtype=np.uint8)
    dfs_cases = [dict(transitive=True, dfs=True, max_depth=3, min_transitive_len=30), dict(transitive=True, dfs=True, max_depth=0, min_transitive_len=150),
                 dict(transitive=True, dfs=True, max_depth=2, min_transitive_len=10, min_distance_between_ranges=0),
                 dict(transitive=True, dfs=True, max_depth=4, min_transitive_len=60, min_distance_between_ranges=40, min_output_length=80),
                 dict(transitive=True, dfs=True, max_depth=2, min_identity=0.8)]
    bfs_cases = [dict(transitive=True, max_depth=3, min_transitive_len=30), dict(transitive=True, max_depth=0, min_transitive_len=200),
                 dict(transitive=True, max_depth=2, min_identity=0.7, min_output_length=50)]
    ref = {}
    for walk in (0, 1, 2):
        g.set_option("walk_kernel", walk)
        for k, kw in enumerate(dfs_cases):
            res = assert_same(g, c, ranges, **kw)
            st, cnt, ck = g.query_batch_stats(ranges, impg_amd.make_params(**kw))
            assert [int(x) for x in cnt] == [len(res[i]) - 1 for i in range(len(ranges))] or kw.get("min_output_length")  # (self rows are not hits)
            if walk == 0:
                ref[("d", k)] = (cnt.tolist(), ck.tolist(), st.projected)
            else:
                assert (cnt.tolist(), ck.tolist(), st.projected) == ref[("d", k)], (walk, kw)
        assert_same(g, c, ranges[:70], subset_keep=keep, transitive=True, dfs=True, max_depth=3, min_transitive_len=30)
        for k, kw in enumerate(bfs_cases):
            for lo in (0, 60):
                assert_same(g, c, ranges[lo:lo + 50], **kw)
            st, cnt, ck = g.query_batch_stats(ranges[:64], impg_amd.make_params(**kw))
            if walk == 0:
                ref[("b", k)] = (cnt.tolist(), ck.tolist(), st.projected)
            else:
                assert (cnt.tolist(), ck.tolist(), st.projected) == ref[("b", k)], (walk, kw)
        assert_same(g, c, ranges[:40], subset_keep=keep, transitive=True, max_depth=3, min_transitive_len=30)
        assert g.query_transitive_dfs(*ranges[0], max_depth=3).tolist() == c.query(*ranges[0], transitive=True, dfs=True, max_depth=3).tolist()
        assert g.query_transitive_bfs(*ranges[0], max_depth=3).tolist() == c.query(*ranges[0], transitive=True, max_depth=3).tolist()


@pytest.mark.parametrize("seed", [21, 22])
def test_update_by_segments_matches_the_library_sort(tmp_path, seed):
    """The visited update's hits grouped query by query (seg_group_kernel: runs in frontier order, a counting sort by
    sequence inside a query -- the default) against the same hits ordered by the library's stable radix sort (option
    segment_groups = 0), and both against the oracle: rows, counts, checksums, BFS and DFS, with masks and a subset
    filter, many sequences, dense targets, ranges without hits."""
    text, _ = random_paf(500 + seed, 700, n_seq=(9 if seed == 21 else 70), seq_len=30_000, max_ops=200, weird=True, self_aln=True)
    g, c = both(tmp_path, text)
    n_seq = g.num_seqs()
    ranges = random_ranges(seed, 300, n_seq, 30_000, max_len=6000, min_len=1)
    g.set_option("walk_kernel", 0)  # (everything on the batch engine, whose update this is)
    cases = [dict(transitive=True, max_depth=3, min_transitive_len=20), dict(transitive=True, max_depth=0, min_transitive_len=150),
             dict(transitive=True, max_depth=4, min_transitive_len=40, min_distance_between_ranges=30, min_output_length=60),
             dict(transitive=True, dfs=True, max_depth=3, min_transitive_len=30)]
    mask = random_mask(seed, n_seq, 30_000, present=0.7)
    for kw in cases:
        out = {}
        for seg in (1, 0):
            g.set_option("segment_groups", seg)
            st, cnt, ck = g.query_batch_stats(ranges, impg_amd.make_params(**kw))
            out[seg] = (st.projected, cnt.tolist(), ck.tolist())
            if not kw.get("dfs") or seg == 1:
                assert_same(g, c, ranges[:120], **kw)
                assert_same(g, c, ranges[100:160], masked_regions=mask, **kw)
        assert out[0] == out[1], kw
        # ... and a query's hits cut into slices of its frontier ranges (what a level with more than a wave's worth of hits
        # per query gets; forced here on every level, more slices than some queries have ranges)
        g.set_option("segment_groups", 1)
        for parts in (2, 5, 64):
            g.set_option("segment_parts", parts)
            st, cnt, ck = g.query_batch_stats(ranges, impg_amd.make_params(**kw))
            assert (st.projected, cnt.tolist(), ck.tolist()) == out[1], (kw, parts)
        assert_same(g, c, ranges[:120], **kw)
        assert_same(g, c, ranges[100:160], masked_regions=mask, **kw)
        g.set_option("segment_parts", 0)
    g.set_option("segment_groups", 1)
    g.set_option("chunk_ranges", 37)  # (chunks: a query's index inside its chunk is what the keys carry)
    assert_same(g, c, ranges, transitive=True, max_depth=3, min_transitive_len=20)


@pytest.mark.parametrize("seed", [11, 12])
def test_walk_grid_form_and_masks(tmp_path, seed):
    """The walk's grid form (a depth-limited BFS of <= 64 ranges: `walk_members` workgroups per query share the last level
    out, walk_final_level) and masked_regions inside the walk (the shape partition.rs:359-391 calls, BFS and DFS): rows
    against the oracle, counts / checksums / projections against the batch engine, for 1 .. many members, batches of 1,
    3 and 64 ranges, depth 1 .. 4, the length / distance / identity / subset cut-offs -- and the counters say the walk
    answered."""
    text, _ = random_paf(900 + seed, 600, n_seq=7, seq_len=40_000, max_ops=300, weird=True, inconsistent=(seed == 12), self_aln=True)
    g, c = both(tmp_path, text)
    ranges = random_ranges(seed, 128, 7, 40_000, max_len=5000, min_len=1)
    keep = np.array([1, 0, 1, 1, 0, 1, 1], dtype=np.uint8)
    bfs_cases = [dict(transitive=True, max_depth=3, min_transitive_len=30), dict(transitive=True, max_depth=2),
                 dict(transitive=True, max_depth=4, min_transitive_len=60, min_distance_between_ranges=40, min_output_length=80),
                 dict(transitive=True, max_depth=2, min_identity=0.7, min_output_length=50)]
    g.set_option("walk_kernel", 0)
    ref = [g.query_batch_stats(ranges[:64], impg_amd.make_params(**kw)) for kw in bfs_cases]
    g.set_option("walk_kernel", 1)
    for members in (0, 2, 5, 64, 1):
        g.set_option("walk_members", members)
        before = g.counter("walk_launches")
        for k, kw in enumerate(bfs_cases):
            for lo, n in ((0, 64), (64, 3), (70, 1), (90, 38)):
                assert_same(g, c, ranges[lo:lo + n], **kw)
            st, cnt, ck = g.query_batch_stats(ranges[:64], impg_amd.make_params(**kw))
            assert (cnt.tolist(), ck.tolist(), st.projected) == (ref[k][1].tolist(), ref[k][2].tolist(), ref[k][0].projected), (members, kw)
        assert_same(g, c, ranges[:40], subset_keep=keep, transitive=True, max_depth=3, min_transitive_len=30)
        assert g.query_transitive_bfs(*ranges[5], max_depth=3).tolist() == c.query(*ranges[5], transitive=True, max_depth=3).tolist()
        if members != 1:
            assert g.counter("walk_launches") > before and g.counter("walk_fallbacks") == 0
            assert g.counter("walk_members") == {0: 32, 2: 2, 5: 5, 64: 64}[members]  # (the last call: one range)
        else:
            assert g.counter("walk_launches") == before  # (no grid form: the batch engine keeps a BFS)
    # masks: the walk (grid BFS, DFS batches of any size, every small BFS under walk_kernel = 2) and the batch engine
    g.set_option("walk_members", 0)
    for m_seed, present, odd in ((1, 1.0, False), (2, 0.6, True), (3, 0.3, True)):
        mask = random_mask(seed * 10 + m_seed, 7, 40_000, present=present, odd_lengths=odd)
        for walk in (1, 2, 0):
            g.set_option("walk_kernel", walk)
            before = g.counter("walk_launches")
            for kw in [dict(transitive=True, max_depth=2), dict(transitive=True, max_depth=3, min_transitive_len=20, min_distance_between_ranges=10),
                       dict(transitive=True, dfs=True, max_depth=3, min_transitive_len=20, min_distance_between_ranges=10),
                       dict(transitive=True, max_depth=0, min_transitive_len=300, min_distance_between_ranges=10),
                       dict(transitive=True, dfs=True, max_depth=0, min_transitive_len=300)]:
                assert_same(g, c, ranges[:48], masked_regions=mask, **kw)
                assert_same(g, c, ranges[100:101], masked_regions=mask, **kw)
            if kw.get("dfs"):
                assert_same(g, c, ranges, masked_regions=mask, **kw)  # a DFS batch of any size
            assert (g.counter("walk_launches") > before) == (walk != 0)
            assert g.counter("walk_fallbacks") == 0
    # a mask with empty ranges: its self pieces touch; the DFS goes to the batch engine, the BFS stays
    g.set_option("walk_kernel", 1)
    L = int(c.seq_len(int(ranges[0][0])))
    t0 = int(ranges[0][0])
    mask = {t0: (L, [(1000, 1000), (1500, 1500), (2500, 2600)])}
    probe = [(t0, 400, 3000), (t0, 1000, 1500), (t0, 0, L)]
    for kw in [dict(transitive=True, dfs=True, max_depth=3, min_transitive_len=20), dict(transitive=True, max_depth=3, min_transitive_len=20),
               dict(transitive=True, dfs=True, max_depth=2, min_transitive_len=700)]:
        assert_same(g, c, probe, masked_regions=mask, **kw)


# ---- the row stream: impg_gpu_query_batch_stream ---------------------------------
def test_query_batch_stream_matches_the_one_shot_call(tmp_path):
    """Chunks arrive in range order, one at a time, and hold the rows (and CIGARs) impg_gpu_query_batch returns for the same
    ranges -- whatever the chunk size, also when chunks are halved to stay under the block size, under a mask / a subset
    filter, and a consumer can stop the stream (main.rs:7435-7470 is the loop this feeds)."""
    text, _ = random_paf(4242, 300, n_seq=7, seq_len=30000, self_aln=True, max_ops=120)
    g, c = both(tmp_path, text)
    ranges = random_ranges(17, 61, g.num_seqs(), 30000, max_len=4000, min_len=100)
    seq_len = int(c.seq_len(0))
    mask = {0: (seq_len, [(100, 2000), (5000, 9000)]), 2: (seq_len, [(0, 700)])}
    keep = np.array([1, 0, 1, 1, 0, 1, 1], dtype=np.uint8)
    cases = [(dict(), {}), (dict(transitive=True, max_depth=3, min_transitive_len=20), {}),
             (dict(transitive=True, dfs=True, max_depth=2, min_transitive_len=40), {}),
             (dict(transitive=True, max_depth=2, multi_impg=True), {}),
             (dict(transitive=True, max_depth=2), dict(masked_regions=mask)),
             (dict(transitive=True, max_depth=2), dict(subset_keep=keep)),
             (dict(transitive=True, max_depth=2, min_transitive_len=40, store_cigar=True), {})]
    for kw, extra in cases:
        p = impg_amd.make_params(**kw)
        want = g.query_batch(ranges, p, **extra)
        for chunk, block in ((7, 0), (1, 0), (1000, 0), (16, 24 * 10)):
            seen = []

            def consume(first, part):
                assert first == (seen[-1][0] + seen[-1][1] if seen else 0)  # in range order, nothing skipped
                rows = [part[i].tolist() for i in range(len(part))]
                cg = [[x.tolist() for x in part.cigars(i)] for i in range(len(part))] if kw.get("store_cigar") else None
                seen.append((first, len(part), rows, cg))
                return False

            proj = g.query_batch_stream(ranges, consume, p, chunk_ranges=chunk, max_block_bytes=block, **extra)
            assert proj == want.projected, (kw, chunk)
            assert sum(n for _, n, _, _ in seen) == len(ranges)
            if block:  # (chunks of several ranges are halved until their rows fit the block: 10 rows here)
                assert all(n == 1 or sum(len(r) for r in rows) <= 10 for _, n, rows, _ in seen) and len(seen) > len(ranges) // 16 + 1
            i = 0
            for first, n, rows, cg in seen:
                for k in range(n):
                    assert rows[k] == want[i].tolist(), (kw, chunk, i)
                    if cg is not None:
                        assert cg[k] == [x.tolist() for x in want.cigars(i)], (kw, chunk, i)
                    i += 1
    # a consumer that stops after the second chunk; one that raises
    calls = []
    g.query_batch_stream(ranges, lambda first, part: calls.append(first) or len(calls) >= 2, impg_amd.make_params(transitive=True), chunk_ranges=5)
    assert calls == [0, 5]
    with pytest.raises(ZeroDivisionError):
        g.query_batch_stream(ranges, lambda first, part: 1 // 0, impg_amd.make_params(), chunk_ranges=5)
    assert g.query_batch_stream([], lambda first, part: False, impg_amd.make_params()) == 0
    # the handle is fine afterwards
    assert_same(g, c, ranges[:10], transitive=True, max_depth=2)


def test_prewarm_options(tmp_path):
    """"prewarm_result_bytes" / "prewarm_walk": the first call's one-off allocations made ahead of it; results unchanged."""
    text, _ = random_paf(99, 200, n_seq=5, seq_len=20000)
    g, c = both(tmp_path, text)
    g.set_option("prewarm_result_bytes", 8 << 20)
    g.set_option("prewarm_walk", 2)
    ranges = random_ranges(3, 20, g.num_seqs(), 20000, max_len=3000, min_len=100)
    assert_same(g, c, ranges, transitive=True, max_depth=3, min_transitive_len=20)
    assert_same(g, c, ranges, transitive=True, dfs=True, max_depth=2, min_transitive_len=20)
    assert g.query_transitive_bfs(*ranges[0], max_depth=2).tolist() == c.query(*ranges[0], transitive=True, max_depth=2).tolist()
    with pytest.raises(impg_amd.ImpgGpuError):
        g.set_option("prewarm_walk", 3)


def test_lookup_order_sort():
    """The hand-written stable argsort behind a level's lookup order (order_scatter_kernel): a permutation, keys
    non-decreasing by their low bits, equal keys in index order -- checked on the host by impg_gpu_selftest_order_sort for
    sizes around the tile (8 192 keys) and pass boundaries (digits of <= 8 bits: 1 .. 4 passes), repeated keys, and bits
    above end_bit set."""
    from impg_amd import _lib
    for n, bits in [(1, 1), (2, 1), (63, 3), (64, 7), (8191, 8), (8192, 9), (8193, 16), (100_000, 17), (1_000_003, 21), (3_000_000, 24),
                    (500_000, 25), (70_000, 32), (2_000_000, 5), (16_384, 21)]:
        for seed in (1, 2):
            _lib.check(_lib.lib().impg_gpu_selftest_order_sort(0, n, bits, seed + 10 * bits))


def test_update_slices_a_huge_query(tmp_path):
    """The visited update's automatic choices (no option forced).  An index of a dense component (s0-s2: every level of a
    whole-sequence query holds tens of thousands of hits) and a sparse one (t0-t2).  A batch of small queries on the sparse
    component with two whole-sequence queries on the dense one: the level's average is small, one query holds more hits
    than a wave should take, so the level is counted again in slices cut for it (segment_retries); then a batch of
    whole-sequence queries only (sliced from the start).  Counts, checksums and projections equal the library-sort form's;
    rows equal the oracle's on the small queries and on one whole-sequence query."""
    dense, _ = random_paf(4242, 45_000, n_seq=3, seq_len=40_000, max_ops=40, weird=False, self_aln=False)
    sparse, _ = random_paf(4243, 400, n_seq=3, seq_len=40_000, max_ops=40, weird=False, self_aln=False)
    sparse = "\n".join(ln.replace("s0\t", "t0\t").replace("s1\t", "t1\t").replace("s2\t", "t2\t") for ln in sparse.splitlines()) + "\n"
    g, c = both(tmp_path, dense + sparse)
    g.set_option("walk_kernel", 0)
    sid = {g.seq_name(i): i for i in range(g.num_seqs())}
    small = [(sid["t%d" % t], a, b) for t, a, b in random_ranges(5, 60, 3, 40_000, max_len=3000, min_len=200)]
    whole = [(sid["s0"], 0, 40_000), (sid["s1"], 0, 40_000), (sid["s2"], 100, 39_000)]
    kw = dict(transitive=True, max_depth=3, min_transitive_len=10)
    p = impg_amd.make_params(**kw)
    for batch, want_retry in ((small + whole[:2], True), (whole, False)):
        before = {k: g.counter(k) for k in ("segment_sliced_levels", "segment_retries", "segment_library_levels")}
        g.set_option("segment_groups", 1)
        st1, cnt1, ck1 = g.query_batch_stats(batch, p)
        after = {k: g.counter(k) for k in before}
        assert after["segment_sliced_levels"] > before["segment_sliced_levels"], (before, after)
        if want_retry:
            assert after["segment_retries"] > before["segment_retries"], (before, after)
        assert after["segment_library_levels"] == before["segment_library_levels"], (before, after)
        g.set_option("segment_groups", 0)
        st0, cnt0, ck0 = g.query_batch_stats(batch, p)
        assert (st1.projected, cnt1.tolist(), ck1.tolist()) == (st0.projected, cnt0.tolist(), ck0.tolist())
        assert int(cnt1.max()) > 32768
    g.set_option("segment_groups", 1)
    assert_same(g, c, small[:12] + whole[:1], **kw)
