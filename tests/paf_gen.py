"""Random PAF text for parity tests (inputs only; no reference code)."""
import numpy as np

OPS = "=XIDM"


def random_cigar(rng, n_ops, weird=False):
    ops = []
    for _ in range(n_ops):
        c = OPS[rng.integers(0, 5)] if rng.random() < 0.5 else "="
        if weird and rng.random() < 0.08:
            ln = 0  # zero-length op: hits the (0, query_delta) arm whatever its letter
        else:
            ln = int(rng.integers(1, 40)) if rng.random() < 0.9 else int(rng.integers(40, 400))
        ops.append((ln, c))
    return ops


def spans(ops):
    td = sum(l for l, c in ops if c in "=XDM")
    qd = sum(l for l, c in ops if c in "=XIM")
    return td, qd


def random_paf(seed, n_records, n_seq=6, seq_len=20000, max_ops=150, weird=False, inconsistent=False, self_aln=False):
    """Returns (text, names).  Many records over few sequences so ranges hit
    several alignments; CIGAR lengths straddle the 32-op tile boundaries."""
    rng = np.random.default_rng(seed)
    names = ["s%d" % i for i in range(n_seq)]
    lines = []
    for _ in range(n_records):
        n_ops = int(rng.choice([1, 2, 3, 27, 28, 29, 31, 32, 33, 55, 56, 57, 63, 64, 65, 96, 223, 224, 225,
                                int(rng.integers(1, max_ops + 1)), int(rng.integers(1, max_ops + 1))]))
        n_ops = min(n_ops, max(max_ops, 3))
        ops = random_cigar(rng, n_ops, weird)
        td, qd = spans(ops)
        if td == 0 or qd == 0:
            ops.append((int(rng.integers(1, 30)), "="))
            td, qd = spans(ops)
        if td >= seq_len or qd >= seq_len:
            continue
        t = int(rng.integers(0, n_seq))
        q = int(rng.integers(0, n_seq))
        if not self_aln and q == t:
            q = (t + 1) % n_seq
        ts = int(rng.integers(0, seq_len - td))
        qs = int(rng.integers(0, seq_len - qd))
        te, qe = ts + td, qs + qd
        if inconsistent and rng.random() < 0.3:  # coordinates that disagree with the CIGAR
            te = max(ts + 1, te + int(rng.integers(-20, 21)))
            qe = max(qs + 1, qe + int(rng.integers(-20, 21)))
            te, qe = min(te, seq_len), min(qe, seq_len)
        strand = "+-"[rng.integers(0, 2)]
        cg = "".join("%d%s" % (l, c) for l, c in ops)
        lines.append("%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t60\tcg:Z:%s" %
                     (names[q], seq_len, qs, qe, strand, names[t], seq_len, ts, te, td, td + qd, cg))
    return "\n".join(lines) + "\n", names


def random_ranges(seed, n, n_seq, seq_len, max_len=3000, min_len=1):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        t = int(rng.integers(0, n_seq))
        ln = int(rng.integers(min_len, max_len + 1))
        s = int(rng.integers(0, seq_len - ln + 1))
        out.append((t, s, s + ln))
    return out
