"""Parity at BASELINE.json's full index size (1M-record PAF, -x -m 3): the oracle
cannot run the whole batch, so (a) a sample of ranges is compared exactly through
per-range (count, order-independent checksum) pairs computed on the device, and
(b) size-independent properties are checked on the full batch: results do not
depend on how the batch is chunked, and repeated runs are identical."""
import os

import numpy as np
import pytest

import impg_amd
from oracle import oracle as o

pytestmark = pytest.mark.gpu

M1 = np.uint64(0xBF58476D1CE4E5B9)
M2 = np.uint64(0x94D049BB133111EB)


def mix64(z):
    z = (z ^ (z >> np.uint64(30))) * M1
    z = (z ^ (z >> np.uint64(27))) * M2
    return z ^ (z >> np.uint64(31))


def checksum(res):
    """hit_stats_kernel's per-range checksum, from oracle results (self interval excluded)."""
    with np.errstate(over="ignore"):
        u = lambda a: a.astype(np.int64).astype(np.uint64) & np.uint64(0xFFFFFFFF)
        a = mix64((u(res["query_id"]) << np.uint64(32)) | u(res["q_first"]))
        a = mix64(a ^ ((u(res["q_last"]) << np.uint64(32)) | u(res["target_id"])))
        a = mix64(a ^ ((u(res["t_first"]) << np.uint64(32)) | u(res["t_last"])))
        return int(a.sum(dtype=np.uint64))


@pytest.fixture(scope="module")
def full(tmp_path_factory):
    d = tmp_path_factory.mktemp("full")
    paf = str(d / "synth_1m.paf")
    impg_amd.synth_paf_text(paf, 42, 1_000_000)
    g = impg_amd.GpuImpg.from_paf(paf)
    bed = impg_amd.synth_bed(7, 4096)
    ranges = np.zeros(len(bed), dtype=impg_amd.RANGE_DTYPE)
    ranges["target_id"] = [g.seq_id(impg_amd.synth_seq_name(int(t))) for t in bed["target_id"]]
    ranges["start"], ranges["end"] = bed["start"], bed["end"]
    return paf, g, ranges


def test_headline_config_sample_vs_oracle(full):
    paf, g, ranges = full
    p = impg_amd.make_params(transitive=True, max_depth=3)
    g.set_option("chunk_ranges", 2048)
    st, cnt, ck = g.query_batch_stats(ranges, p)
    assert st.levels == 3 and st.projected > 10_000 * len(ranges)
    c = o.OracleIndex(paf_paths=[paf], preparse=False)  # pread + parse per hit, like the reference
    rng = np.random.default_rng(1)
    sample = sorted(set(rng.integers(0, len(ranges), 70).tolist()) | {0, len(ranges) - 1})  # (>= 64 ranges: ~1.4e6 projections)
    total = 0
    for i in sample:
        r = ranges[i]
        want = c.query(int(r["target_id"]), int(r["start"]), int(r["end"]), transitive=True, max_depth=3)
        hits = want[1:]  # one self interval first (the range lies inside the sequence)
        assert int(cnt[i]) == len(hits), i
        assert int(ck[i]) == checksum(hits), i
        total += len(hits)
    assert len(sample) >= 64 and total > 1_000_000  # the sample itself is a seven-figure number of projections


def test_chunking_and_repeat_invariance(full):
    paf, g, ranges = full
    p = impg_amd.make_params(transitive=True, max_depth=3)
    g.set_option("chunk_ranges", 4096)
    st1, cnt1, ck1 = g.query_batch_stats(ranges, p)
    g.set_option("chunk_ranges", 333)  # ragged chunks
    st2, cnt2, ck2 = g.query_batch_stats(ranges, p)
    g.set_option("pair_budget", 1 << 22)  # force automatic splitting inside a chunk
    g.set_option("chunk_ranges", 4096)
    st3, cnt3, ck3 = g.query_batch_stats(ranges, p)
    g.set_option("pair_budget", 1 << 29)
    assert st1.projected == st2.projected == st3.projected == int(cnt1.sum())
    assert (cnt1 == cnt2).all() and (ck1 == ck2).all()
    assert (cnt1 == cnt3).all() and (ck1 == ck3).all()


def test_nontransitive_full_results_sample(full):
    """BASELINE config 2 (no transitive): full results through the C ABI for a
    slice of the batch, every interval compared with the oracle."""
    paf, g, ranges = full
    c = o.OracleIndex(paf_paths=[paf], preparse=False)
    sub = ranges[:200]
    res = g.query_batch(sub, impg_amd.make_params())
    for i in range(200):  # all of them
        r = sub[i]
        want = c.query(int(r["target_id"]), int(r["start"]), int(r["end"]))
        assert res[i].tolist() == want.tolist()
    assert res.projected == sum(len(res[i]) - 1 for i in range(200))


def test_headline_batch_prefix_consistency(full):
    """The headline batch itself (100 000 ranges, -x -m 3): queries are independent, so the first 4 096 ranges of the
    full batch must give exactly the per-range counts and checksums the 4 096-range batch gives (which the oracle
    sample above pins), whatever chunks, lookup orders and slot layouts the big batch runs through; and the whole
    batch's total is the sum of its per-range counts."""
    paf, g, ranges = full
    bed = impg_amd.synth_bed(7, 100_000)
    big = np.zeros(len(bed), dtype=impg_amd.RANGE_DTYPE)
    big["target_id"] = [g.seq_id(impg_amd.synth_seq_name(int(t))) for t in bed["target_id"]]
    big["start"], big["end"] = bed["start"], bed["end"]
    assert (big[:len(ranges)] == ranges).all()  # (the generator is a prefix-stable stream)
    p = impg_amd.make_params(transitive=True, max_depth=3)
    g.set_option("chunk_ranges", 2048)
    st0, cnt0, ck0 = g.query_batch_stats(ranges, p)
    g.set_option("chunk_ranges", 50000)
    g.set_option("pair_budget", 1 << 30)
    st, cnt, ck = g.query_batch_stats(big, p)
    g.set_option("pair_budget", 1 << 29)
    assert (cnt[:len(ranges)] == cnt0).all() and (ck[:len(ranges)] == ck0).all()
    assert st.projected == int(cnt.sum()) and st.projected > 2_000_000_000
