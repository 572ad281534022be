#!/usr/bin/env python3
"""BASELINE config 1 stand-in: the cerevisiae PAF the reference's README queries
(`impg query -r S288C#1#chrI:50000-100000 -d 1000`) is not in the reference tree, so this writes a seeded 96-record
PAF with PanSN yeast-style names -- seven chrI assemblies, all-vs-all style, both strands, =/X/I/D CIGARs -- and
the BED text the oracle prints for that command line (plain and `-x -m 2`).  tests/test_golden.py compares the
oracle (CPU, drift guard) and the engine's CLI (GPU) with the committed text byte for byte.

usage: python tests/golden/make_config1.py        (from the repo root; rewrites config1_yeast.paf / config1_expected.json)"""
import json, os, random, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as o

STRAINS = [("S288C", 230218), ("DBVPG6044", 224126), ("Y12", 218964), ("SK1", 228861), ("YPS128", 222107),
           ("UWOPS034614", 216482), ("DBVPG6765", 225590)]
NAMES = ["%s#1#chrI" % s for s, _ in STRAINS]
LENS = dict(zip(NAMES, [n for _, n in STRAINS]))


def cigar(rng, t_span):
    """ops summing to t_span on the target: '=' runs of 20..400 separated by X (60 %), I / D of 1..30 (20 % each)"""
    ops, t, q = [], 0, 0
    while t < t_span:
        run = min(rng.randint(20, 400), t_span - t)
        ops.append("%d=" % run); t += run; q += run
        if t >= t_span:
            break
        r = rng.random()
        if r < 0.6:
            n = min(rng.randint(1, 3), t_span - t)
            ops.append("%dX" % n); t += n; q += n
        elif r < 0.8:
            n = rng.randint(1, 30)
            ops.append("%dI" % n); q += n
        else:
            n = min(rng.randint(1, 30), t_span - t)
            ops.append("%dD" % n); t += n
    return "".join(ops), q


def main():
    rng = random.Random(20260929)
    lines = []
    for i in range(96):
        t = NAMES[0] if i % 3 == 0 else rng.choice(NAMES)  # a third of the records have S288C#1#chrI as target
        qn = rng.choice([n for n in NAMES if n != t])
        t_span = rng.randint(4000, 45000)
        # S288C targets are placed so that most of them touch the queried window 50000-100000
        lo, hi = (20000, 110000) if t == NAMES[0] else (0, LENS[t] - t_span)
        ts = rng.randint(lo, min(hi, LENS[t] - t_span))
        cg, q_span = cigar(rng, t_span)
        qs = rng.randint(0, LENS[qn] - q_span)
        strand = "+" if rng.random() < 0.6 else "-"
        matches = sum(int(x) for x in __import__("re").findall(r"(\d+)=", cg))
        lines.append("\t".join(map(str, [qn, LENS[qn], qs, qs + q_span, strand, t, LENS[t], ts, ts + t_span, matches, max(t_span, q_span), 60,
                                         "tp:A:P", "cg:Z:" + cg])))
    paf = os.path.join(HERE, "config1_yeast.paf")
    with open(paf, "w") as f:
        f.write("\n".join(lines) + "\n")
    ix = o.OracleIndex(paf_paths=[paf], preparse=True)
    exp = {"command": "impg query -a config1_yeast.paf -r S288C#1#chrI:50000-100000 -d 1000 [-x -m 2]",
           "bed": ix.query_bed("S288C#1#chrI", 50000, 100000, merge_distance=1000),
           "bed_transitive_m2": ix.query_bed("S288C#1#chrI", 50000, 100000, merge_distance=1000, transitive=True, max_depth=2)}
    with open(os.path.join(HERE, "config1_expected.json"), "w") as f:
        json.dump(exp, f, indent=1)
    print(len(lines), "records;", exp["bed"].count("\n"), "BED rows,", exp["bed_transitive_m2"].count("\n"), "with -x -m 2")


if __name__ == "__main__":
    main()
