"""Random tracepoint alignments for the approximate-mode tests (inputs only)."""
import numpy as np

from oracle import oracle as o


def random_tp(seed, n_records, n_seq=5, seq_len=200_000, fastga=False, trace_spacing=100, max_segs=60, self_aln=False):
    """-> dict for OracleIndex(tracepoints=...) / GpuImpg.from_tracepoints (same arrays for both)."""
    rng = np.random.default_rng(seed)
    rec = np.zeros(n_records, dtype=o.TP_RECORD_DTYPE)
    tps, qds, dfs = [], [], []
    off = 0
    for i in range(n_records):
        n = int(rng.choice([1, 2, 3, int(rng.integers(1, max_segs + 1))]))
        tp = rng.integers(0, 260, n)
        tp[rng.random(n) < 0.08] = 0  # pure insertions in the query
        if fastga:
            qcs = int(rng.integers(0, 5000))
            fb = ((qcs // trace_spacing) + 1) * trace_spacing - qcs
            qd = np.full(n, trace_spacing)
            qd[0] = fb
            df = rng.integers(0, 30, n)
        else:
            qcs = 0
            qd = rng.integers(0, 260, n)
            qd[rng.random(n) < 0.08] = 0  # pure deletions
            df = np.zeros(n, dtype=np.int64)
        tspan, qspan = int(tp.sum()), int(qd.sum())
        if tspan == 0 or qspan == 0:
            tp[0] += 50
            if not fastga:
                qd[0] += 50
            tspan, qspan = int(tp.sum()), int(qd.sum())
        t = int(rng.integers(0, n_seq))
        q = int(rng.integers(0, n_seq))
        if not self_aln and q == t:
            q = (t + 1) % n_seq
        ts = int(rng.integers(0, seq_len - tspan))
        qs = int(rng.integers(0, seq_len - qspan))
        rec[i] = (q, t, qs, qs + qspan, ts, ts + tspan, off, n, int(rng.integers(0, 2)), qcs)
        tps.append(tp); qds.append(qd); dfs.append(df)
        off += n
    d = dict(records=rec, tracepoints=np.concatenate(tps).astype(np.int32), seq_len=np.full(n_seq, seq_len, dtype=np.int64),
             fastga=fastga, trace_spacing=trace_spacing if fastga else 0, max_complexity=0 if fastga else 12)
    d["query_deltas"] = None if fastga else np.concatenate(qds).astype(np.int32)
    d["diffs"] = np.concatenate(dfs).astype(np.int32) if fastga else None
    return d
