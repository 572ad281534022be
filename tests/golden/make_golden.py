#!/usr/bin/env python3
"""Regenerates tests/golden/golden_*.json.gz.

Inputs: the five PAF data files the reference's own tests hold (tests/test_data/crush/**.paf in the reference
tree, copied unchanged to tests/golden/ref_paf/ -- real aligner output: =/X/I/D CIGARs, self-similar fragment
sets, both strands) and one seeded synthetic PAF (tests/paf_gen.py).  Expected outputs: the CPU oracle
(oracle/, pinned to the reference's known-answer tests by tests/test_oracle_kat.py) on a fixed list of
queries per file.  The reference itself cannot be run in this image (no Rust toolchain), so these files pin the
oracle against drift and give the GPU tests something committed to compare with; they are not reference runs.

usage: python tests/golden/make_golden.py          (from the repo root; rewrites the JSON files)"""
import gzip, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as o
from tests.paf_gen import random_paf

CASES = [  # name -> params
    ("query", dict()),
    ("query_identity", dict(min_identity=0.95)),
    ("bfs_default", dict(transitive=True, max_depth=2)),
    ("bfs_deep", dict(transitive=True, max_depth=0, min_transitive_len=30, min_distance_between_ranges=5)),
    ("dfs", dict(transitive=True, dfs=True, max_depth=3, min_transitive_len=50, min_distance_between_ranges=10,
                 min_output_length=40)),
    ("multi_bfs", dict(transitive=True, max_depth=2, min_transitive_len=50, multi_impg=True)),
]


def ranges_for(ix):
    out = []
    for t in range(ix.num_seqs()):
        n = int(ix.seq_len(t))
        if n < 40:
            continue
        out += [(t, 0, n), (t, n // 4, 3 * n // 4), (t, n // 3, n // 3 + 25), (t, n - 60, n)]
    return out[:32]


def mask_for(ix):
    m = {}
    for t in range(0, ix.num_seqs(), 2):  # every other sequence is absent from the map
        n = int(ix.seq_len(t))
        m[t] = (n, [(n // 5, n // 5 + 20), (n // 2, n // 2 + 15)] if n > 120 else [])
    return m


def build(name, paf_path, text=None):
    if text is not None:
        ix = o.OracleIndex(paf_text=text)
    else:
        ix = o.OracleIndex(paf_paths=[paf_path], preparse=True)
    ranges = ranges_for(ix)
    doc = {"name": name, "paf": os.path.relpath(paf_path, HERE) if text is None else None, "paf_text": text,
           "seqs": [[ix.seq_name(i), int(ix.seq_len(i))] for i in range(ix.num_seqs())],
           "ranges": ranges, "cases": []}
    for cname, kw in CASES:
        rows, proj = [], 0
        for (t, s, e) in ranges:
            rows.append([[int(x) for x in r] for r in ix.query(t, s, e, **kw).tolist()])
            proj += ix.last_projection_count()
        doc["cases"].append({"case": cname, "params": kw, "rows": rows, "projected": proj})
    # masked_regions
    mask = mask_for(ix)
    kw = dict(transitive=True, max_depth=3, min_transitive_len=20, min_distance_between_ranges=0)
    rows = [[[int(x) for x in r] for r in ix.query(t, s, e, masked_regions=mask, **kw).tolist()] for (t, s, e) in ranges]
    doc["masked"] = {"params": kw, "mask": {str(k): [v[0], v[1]] for k, v in mask.items()}, "rows": rows}
    # store_cigar + the three text outputs (ranges long enough for perform_query's validation)
    kw = dict(transitive=True, max_depth=2, min_transitive_len=40)
    sub = [(t, s, e) for (t, s, e) in ranges if e - s >= 40][:12]
    texts = {"ranges": sub, "params": kw, "cigars": [], "bed": [], "paf": [], "bedpe": []}
    for k, (t, s, e) in enumerate(sub):
        res, cg = ix.query_cigar(t, s, e, **kw)
        texts["cigars"].append([[int(v) for v in c.tolist()] for c in cg])
        nm = "%s:%d-%d" % (ix.seq_name(t), s, e)
        texts["bed"].append(ix.query_bed(ix.seq_name(t), s, e, range_name=nm, merge_distance=10, **kw))
        for fmt in ("paf", "bedpe"):
            try:
                texts[fmt].append(ix.query_paf(ix.seq_name(t), s, e, range_name=nm, merge_distance=10, fmt=fmt, **kw))
            except RuntimeError:
                texts[fmt].append(None)  # nothing left after dropping the input range: the reference panics
    doc["texts"] = texts
    with open(os.path.join(HERE, "golden_%s.json.gz" % name), "wb") as f:
        with gzip.GzipFile(fileobj=f, mode="wb", mtime=0) as z:  # (mtime 0: byte-identical when regenerated)
            z.write(json.dumps(doc, separators=(",", ":")).encode())
    n_rows = sum(len(r) for c in doc["cases"] for r in c["rows"])
    print("%-28s %3d seqs %3d ranges %7d rows" % (name, len(doc["seqs"]), len(ranges), n_rows))


if __name__ == "__main__":
    for fn in sorted(os.listdir(os.path.join(HERE, "ref_paf"))):
        if fn.endswith(".paf"):
            build(fn[:-4], os.path.join(HERE, "ref_paf", fn))
    text, _ = random_paf(20260928, 160, n_seq=5, seq_len=9000, max_ops=40, weird=True, self_aln=True)
    build("synthetic_seed20260928", None, text=text)
