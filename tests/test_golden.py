"""Committed golden fixtures (tests/golden/, made by tests/golden/make_golden.py): the PAF data files the
reference's own tests hold plus one seeded synthetic PAF, with the expected rows / CIGARs / BED / PAF / BEDPE
text of a fixed list of queries.  CPU: the oracle still reproduces them (drift guard).  GPU: the engine,
through the C ABI, reproduces them bit for bit without the oracle in the loop."""
import glob, gzip, json, os

import numpy as np
import pytest

import impg_amd
from oracle import oracle as o

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = sorted(glob.glob(os.path.join(HERE, "golden_*.json.gz")))


def load(path):
    with gzip.open(path, "rb") as f:
        return json.loads(f.read().decode())


def paf_path(doc, tmp_path):
    if doc["paf"] is not None:
        return os.path.join(HERE, doc["paf"])
    p = str(tmp_path / "synthetic.paf")
    with open(p, "w") as f:
        f.write(doc["paf_text"])
    return p


def mask_of(doc):
    return {int(k): (v[0], [tuple(r) for r in v[1]]) for k, v in doc["masked"]["mask"].items()}


def test_golden_set_is_complete():
    assert len(FILES) == 6
    assert len(glob.glob(os.path.join(HERE, "ref_paf", "*.paf"))) == 5


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[7:-8] for p in FILES])
def test_oracle_reproduces_golden(tmp_path, path):
    doc = load(path)
    ix = o.OracleIndex(paf_paths=[paf_path(doc, tmp_path)], preparse=True)
    assert [[ix.seq_name(i), int(ix.seq_len(i))] for i in range(ix.num_seqs())] == doc["seqs"]
    ranges = [tuple(r) for r in doc["ranges"]]
    for case in doc["cases"]:
        proj = 0
        for (t, s, e), want in zip(ranges, case["rows"]):
            assert ix.query(t, s, e, **case["params"]).tolist() == [tuple(r) for r in want], (case["case"], t, s, e)
            proj += ix.last_projection_count()
        assert proj == case["projected"]
    m = doc["masked"]
    for (t, s, e), want in zip(ranges, m["rows"]):
        assert ix.query(t, s, e, masked_regions=mask_of(doc), **m["params"]).tolist() == [tuple(r) for r in want]
    tx = doc["texts"]
    for k, (t, s, e) in enumerate(tx["ranges"]):
        nm = "%s:%d-%d" % (ix.seq_name(t), s, e)
        _, cg = ix.query_cigar(t, s, e, **tx["params"])
        assert [c.tolist() for c in cg] == tx["cigars"][k]
        assert ix.query_bed(ix.seq_name(t), s, e, range_name=nm, merge_distance=10, **tx["params"]) == tx["bed"][k]


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[7:-8] for p in FILES])
def test_engine_reproduces_golden(tmp_path, path):
    doc = load(path)
    g = impg_amd.GpuImpg.from_paf(paf_path(doc, tmp_path))
    assert [[g.seq_name(i), int(g.seq_len(i))] for i in range(g.num_seqs())] == doc["seqs"]
    ranges = [tuple(r) for r in doc["ranges"]]
    for case in doc["cases"]:
        res = g.query_batch(ranges, impg_amd.make_params(**case["params"]))
        for i, want in enumerate(case["rows"]):
            assert res[i].tolist() == [tuple(r) for r in want], (case["case"], ranges[i])
        assert res.projected == case["projected"]
    m = doc["masked"]
    res = g.query_batch(ranges, impg_amd.make_params(**m["params"]), masked_regions=mask_of(doc))
    for i, want in enumerate(m["rows"]):
        assert res[i].tolist() == [tuple(r) for r in want], ("masked", ranges[i])
    tx = doc["texts"]
    sub = [tuple(r) for r in tx["ranges"]]
    names = ["%s:%d-%d" % (g.seq_name(t), s, e) for (t, s, e) in sub]
    params = impg_amd.make_params(store_cigar=True, **tx["params"])
    res = g.query_batch(sub, params)
    for k in range(len(sub)):
        assert [c.tolist() for c in res.cigars(k)] == tx["cigars"][k]
    plain = g.query_batch(sub, impg_amd.make_params(**tx["params"]))
    assert plain.bed(names, merge_distance=10, params=impg_amd.make_params(**tx["params"])) == "".join(tx["bed"])
    for fmt in ("paf", "bedpe"):
        ok = [k for k in range(len(sub)) if tx[fmt][k] is not None]
        if len(ok) == len(sub):
            assert res.paf(names, merge_distance=10, params=params, fmt=fmt) == "".join(tx[fmt])
        else:  # a range with nothing left after dropping the input range: the reference panics, both sides raise
            r2 = g.query_batch([sub[k] for k in ok], params)
            assert r2.paf([names[k] for k in ok], merge_distance=10, params=params, fmt=fmt) == "".join(tx[fmt][k] for k in ok)


# ---- BASELINE config 1 stand-in: `impg query -r S288C#1#chrI:50000-100000 -d 1000` on a committed yeast-style PAF -------
CONFIG1_PAF = os.path.join(HERE, "config1_yeast.paf")


def config1_expected():
    with open(os.path.join(HERE, "config1_expected.json")) as f:
        return json.load(f)


def test_config1_oracle_reproduces_expected_bed():
    exp = config1_expected()
    ix = o.OracleIndex(paf_paths=[CONFIG1_PAF], preparse=True)
    assert ix.num_seqs() == 7 and ix.seq_len(ix.seq_id("S288C#1#chrI")) == 230218
    assert ix.query_bed("S288C#1#chrI", 50000, 100000, merge_distance=1000) == exp["bed"]
    assert ix.query_bed("S288C#1#chrI", 50000, 100000, merge_distance=1000, transitive=True, max_depth=2) == exp["bed_transitive_m2"]
    assert exp["bed"].count("\n") >= 10


@pytest.mark.gpu
def test_config1_cli_prints_expected_bed():
    """The command line of BASELINE config 1 through the engine's CLI (`impg-gpu query`, main.rs:4259-4381): stdout is the
    committed BED text, byte for byte; the library calls give the same bytes."""
    import subprocess
    exp = config1_expected()
    cli = os.path.join(os.path.dirname(impg_amd.__file__), "impg-gpu")
    base = [cli, "query", "-a", CONFIG1_PAF, "-r", "S288C#1#chrI:50000-100000", "-d", "1000"]
    r = subprocess.run(base, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == exp["bed"]
    r = subprocess.run(base + ["-x", "-m", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == exp["bed_transitive_m2"]
    g = impg_amd.GpuImpg.from_paf(CONFIG1_PAF)
    rng = [(g.seq_id("S288C#1#chrI"), 50000, 100000)]
    nm = ["S288C#1#chrI:50000-100000"]
    assert g.query_batch_bed(rng, impg_amd.make_params(), merge_distance=1000, range_names=nm) == exp["bed"]
    assert g.query_batch(rng, impg_amd.make_params()).bed(nm, merge_distance=1000, params=impg_amd.make_params()) == exp["bed"]
