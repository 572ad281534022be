"""CPU-side checks of the product's host logic against the oracle: PAF ingest,
visit-rank, BED merge, parsers, and that the C-ABI library loads and exports
every symbol include/impg_gpu.h declares (no GPU compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import impg_amd
from impg_amd import _lib, index as gi
from oracle import oracle as o
from tests.paf_gen import random_paf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "impg_gpu.h")).read()
    declared = set(re.findall(r"\b(impg_(?:gpu|synth)_[a-z0-9_]+)\s*\(", hdr))
    L = C.CDLL(_lib.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert declared == bound, declared ^ bound


def test_no_gpu_fails_loudly():
    L = impg_amd.lib()
    if L.impg_gpu_device_count() > 0:
        pytest.skip("GPU present")
    rec, ops, sl = impg_amd.synth_paf(1, 10, n_seq=4, seq_len=100000, target_span=2000, n_blocks=10)
    with pytest.raises(impg_amd.ImpgGpuError) as e:
        impg_amd.GpuImpg.from_records(rec, ops, sl)
    assert e.value.code == impg_amd.IMPG_E_HIP


def test_product_never_references_the_oracle():
    for d, _, files in os.walk(os.path.join(ROOT, "impg_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h", ".inc")) or f == "Makefile":
                assert "oracle" not in open(os.path.join(d, f)).read().replace("Independent of the oracle", ""), f


def test_parse_cigar_matches_oracle():
    for s in ["10=5I5D", "1=", "", "100M3X2I", "0=5D", "12345678=1X"]:
        assert gi.parse_cigar(s).tolist() == o.parse_cigar(s).tolist()
    with pytest.raises(ValueError):
        gi.parse_cigar("10=5Q")


def test_parse_target_range():
    for s in ["S288C#1#chrI:50000-100000", "a:b:1-2"]:
        assert gi.parse_target_range(s) == o.parse_target_range(s)
    for bad in ["chr1:5-5", "chr1", "chr1:1-2-3", "chr1:x-9"]:
        with pytest.raises(ValueError):
            gi.parse_target_range(bad)
        with pytest.raises(ValueError):
            o.parse_target_range(bad)


def test_synth_text_and_records_agree(tmp_path):
    path = str(tmp_path / "s.paf")
    shape = dict(n_seq=12, seq_len=200000, target_span=3000, n_blocks=20)
    impg_amd.synth_paf_text(path, 42, 300, **shape)
    rec, ops, sl = impg_amd.synth_paf(42, 300, **shape)
    ix = o.OracleIndex(paf_paths=[path], preparse=True)
    assert ix.num_records() == 300
    lines = open(path).read().splitlines()
    for i in [0, 1, 17, 299]:
        f = lines[i].split("\t")
        r = rec[i]
        assert f[0] == impg_amd.synth_seq_name(int(r["query_id"])) and f[5] == impg_amd.synth_seq_name(int(r["target_id"]))
        assert (int(f[2]), int(f[3]), int(f[7]), int(f[8])) == (r["query_start"], r["query_end"], r["target_start"], r["target_end"])
        assert f[4] == "+-"[int(r["strand"])]
        cg = o.parse_cigar(f[12][5:])
        assert cg.tolist() == ops[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["cigar_len"])].tolist()
        # target span exact, query span = sum of query deltas
        assert int(f[8]) - int(f[7]) == 3000
        assert sum(l for l, c in o.ops_to_pairs(cg) if c in "=XI") == int(f[3]) - int(f[2])
        assert sum(l for l, c in o.ops_to_pairs(cg) if c in "=XD") == 3000


@pytest.mark.parametrize("n", list(range(0, 40)) + [63, 64, 65, 100, 255, 256, 257, 1000, 4097, 20000])
def test_visit_rank_matches_oracle_tree(n):
    """The product's closed-form coitrees visit rank == the order in which the
    oracle's restated BasicCOITree visits nodes when everything overlaps."""
    rank = np.zeros(max(n, 1), dtype=np.uint32)
    assert impg_amd.lib().impg_gpu_visit_rank(n, impg_amd.ORDER_COITREES, rank.ctypes.data) == 0
    if n == 0:
        return
    # n alignments on target T with distinct starts; query the whole sequence
    L = 10 * n + 100
    lines = ["q\t%d\t%d\t%d\t+\tT\t%d\t%d\t%d\t5\t5\t60\tcg:Z:5=" % (L, 10 * i, 10 * i + 5, L, 10 * i, 10 * i + 5) for i in range(n)]
    ix = o.OracleIndex(paf_text="\n".join(lines) + "\n", bidirectional=False, preparse=True)
    res = ix.query(ix.seq_id("T"), 0, L)
    visit = (res["t_first"][1:] // 10).astype(np.int64)  # sorted position of each visited node, in visit order
    assert len(visit) == n
    expect = np.zeros(n, dtype=np.uint32)
    expect[visit] = np.arange(n, dtype=np.uint32)
    assert rank[:n].tolist() == expect.tolist()


def test_visit_rank_sorted_policy():
    rank = np.zeros(17, dtype=np.uint32)
    assert impg_amd.lib().impg_gpu_visit_rank(17, impg_amd.ORDER_SORTED, rank.ctypes.data) == 0
    assert rank.tolist() == list(range(17))


@pytest.mark.parametrize("seed", range(30))
def test_bed_merge_matches_oracle(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(0, 60))
    iv = np.zeros(n, dtype=impg_amd.INTERVAL_DTYPE)
    iv["query_id"] = rng.integers(0, 3, n)
    iv["target_id"] = rng.integers(0, 2, n)
    a = rng.integers(0, 2000, n)
    ln = rng.integers(1, 300, n)
    rev = rng.random(n) < 0.4
    iv["q_first"] = np.where(rev, a + ln, a)
    iv["q_last"] = np.where(rev, a, a + ln)
    t = rng.integers(0, 2000, n)
    iv["t_first"] = t
    iv["t_last"] = t + rng.integers(1, 300, n)
    for d in [-1, 0, 10, 100, 1000]:
        for ms in [True, False]:
            got = gi.bed_merge(iv, d, ms)
            want = o.bed_merge(iv.astype(o.INTERVAL_DTYPE), d, ms)
            assert got.tolist() == want.tolist(), (d, ms)


def test_cli_fails_loudly_without_gpu(tmp_path):
    import subprocess
    if impg_amd.lib().impg_gpu_device_count() > 0:
        pytest.skip("GPU present")
    cli = os.path.join(ROOT, "impg_amd", "impg-gpu")
    paf = tmp_path / "t.paf"
    paf.write_text("A\t1000\t0\t100\t+\tB\t1000\t0\t100\t100\t100\t60\tcg:Z:100=\n")
    r = subprocess.run([cli, "query", "-a", str(paf), "-r", "A:0-100", "-d", "0", "--min-transitive-len", "0"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and r.stdout == "" and "no HIP device" in r.stderr
    r = subprocess.run([cli, "query", "-a", str(paf), "-r", "A:0-100"], capture_output=True, text=True)
    assert r.returncode != 0 and "merge-distance is required" in r.stderr


def test_subset_keep_reference_kat_and_differential():
    """impg_gpu_subset_keep (host-only string logic): the reference's own vectors (subset_filter.rs:181-199), then
    random lists / names against the oracle's transcription."""
    import numpy as np
    import impg_amd
    from oracle import oracle as o
    from tests.test_oracle_kat import SUBSET_KAT, SUBSET_LIST
    got, entries = impg_amd.subset_keep(SUBSET_LIST, [n for n, _ in SUBSET_KAT])
    assert got.tolist() == [w for _, w in SUBSET_KAT] and entries == 5
    rng = np.random.default_rng(5)
    parts = ["HG001", "HG002", "NA12", "chr1", "chr2", "s", "x_y", ""]

    def name():
        k = rng.integers(0, 7)
        a, b = parts[rng.integers(0, len(parts))], parts[rng.integers(0, len(parts))]
        d = str(rng.integers(0, 3)) if rng.random() < 0.7 else ""
        base = [a, a + "#" + d + "#" + b, a + "#" + d, a + "_hap" + d + "_v1", a + "_hap" + d, a + "#" + b, a + "##" + b][k]
        if rng.random() < 0.3:
            base += ":%d-%d" % (rng.integers(0, 50), rng.integers(50, 99))
        return base

    for _ in range(300):
        lines = []
        for _ in range(int(rng.integers(0, 6))):
            x = name()
            x = [x, "  " + x + "\t", "# " + x, x + "\r", ""][rng.integers(0, 5)]
            lines.append(x)
        text = "\n".join(lines) + ("\n" if rng.random() < 0.5 else "")
        names = [name() for _ in range(12)] + [" ", ":", "#", "_hap"]
        want, we = o.subset_matches(text, names)
        got, ge = impg_amd.subset_keep(text, names)
        assert got.tolist() == want.tolist() and ge == we, (text, names)


def test_cli_merge_distance_vectors_rejected_before_any_device_work():
    """parse_merge_distance (main.rs:47-55; vectors main.rs:13709-13714): "10kb" and "3g" are errors, and so is a
    query without -d / --no-merge (main.rs:13401-13416).  The CLI refuses them while parsing its arguments."""
    import os, subprocess
    import impg_amd
    cli = os.path.join(os.path.dirname(impg_amd.__file__), "impg-gpu")
    for bad in (["-d", "10kb"], ["-d", "3g"], ["-d", "-5"], ["-d", "abc"]):
        r = subprocess.run([cli, "query", "-a", "x.paf", "-r", "s:1-200"] + bad, capture_output=True, text=True)
        assert r.returncode != 0 and r.stdout == "" and r.stderr.startswith("Error:"), bad
    # numeric options are parsed whole or refused (clap does the same for the reference): a typo must not become
    # 0, which for -m means "unlimited depth"
    for bad in (["-m", "x"], ["-m", "3x"], ["-m", "70000"], ["--min-transitive-len", "foo"], ["-l", "1e3"],
                ["--min-result-identity", "high"], ["--min-distance-between-ranges", "-1"], ["--order", "random"]):
        r = subprocess.run([cli, "query", "-a", "x.paf", "-r", "s:1-200", "-d", "0"] + bad, capture_output=True, text=True)
        assert r.returncode == 2 and r.stdout == "" and "invalid value" in r.stderr, (bad, r.stderr)


def test_parse_subsequence_reference_kat():
    """impg_gpu_parse_subsequence against the reference's vectors (main.rs:13330-13346) and the oracle on odd names."""
    import impg_amd
    from oracle import oracle as o
    from tests.test_oracle_kat import SUBSEQ_KAT
    for name, want in SUBSEQ_KAT:
        assert impg_amd.parse_subsequence(name) == want
    for name in ["a:1-2", "a:b:10-20", "a:+7-9", "a:-7-9", "a:7", "a:-", ":5-6", "a:2147483647-1", "a:2147483648-1", "a:00012-3",
                 "a:1-2:x", "a:1-2:3-4", "a: 1-2", "a:1 -2", "", ":", "-", "a#1#c:5-", "a:5-5-5"]:
        assert impg_amd.parse_subsequence(name) == o.parse_subsequence(name), name
