"""CPU stand-in for the per-rank engine of impg_amd.sharded, built on the oracle.
Test infrastructure: it lets the multi-rank orchestration (sharding by target,
frontier/hit exchange, re-ordering at home) run under gloo without a GPU."""
import numpy as np
import torch

from impg_amd import _lib
from oracle import oracle as o


class OracleBackend:
    def __init__(self, ix, rank, world):
        self.ix, self.rank, self.world = ix, rank, world
        self.visited = {}

    def begin(self, ranges_t, n, params):
        r = ranges_t.numpy()[:n * 12].view(_lib.RANGE_DTYPE)
        self.visited = {}
        front, self_iv = [], []
        for q in range(n):
            t, s, e = int(r[q]["target_id"]), int(r[q]["start"]), int(r[q]["end"])
            sr = o.SortedRanges(int(self.ix.seq_len(t)), 0)
            pieces = sr.insert(s, e)
            self.visited[(q, t)] = sr
            ps, pe = pieces[0] if pieces else (s, s)
            self_iv.append((t, ps, pe, q))
            if pieces and abs(ps - pe) >= params.min_transitive_len:
                front.append((t, ps, pe, q))
        return (torch.tensor(front, dtype=torch.int32).view(-1, 4), torch.tensor(self_iv, dtype=torch.int32).view(-1, 4))

    def expand(self, frontier, transitive, params, want_hits=True, compact=False):
        rows, accepted = [], 0
        f = frontier.numpy()
        for i in range(f.shape[0]):
            t, s, e = int(f[i, 0]), int(f[i, 1]), int(f[i, 2])
            assert t % self.world == self.rank, "record routed to the wrong shard"
            if transitive:
                res = self.ix.query(t, s, e, transitive=True, max_depth=1, min_transitive_len=0, min_distance_between_ranges=0)
            else:
                res = self.ix.query(t, s, e)
            for k, x in enumerate(res[1:]):
                rows.append((i, int(x["query_id"]), int(x["q_first"]), int(x["q_last"]), int(x["t_first"]), int(x["t_last"]), k, 0))
            accepted += len(res) - 1
        h = torch.tensor(rows, dtype=torch.int32).view(-1, 8)
        if compact:
            h = h[:, :4].contiguous()
        return (h if want_hits else h[:0]), accepted

    def update(self, frontier, hits, params):
        f, h = frontier.numpy(), hits.numpy()
        nxt = {}
        for k in range(h.shape[0]):
            fi = int(h[k, 0])
            q, cur_t = int(f[fi, 3]), int(f[fi, 0])
            qid, qs, qe = int(h[k, 1]), int(h[k, 2]), int(h[k, 3])
            if qid == cur_t:
                continue
            sr = self.visited.get((q, qid))
            if sr is None:
                sr = self.visited[(q, qid)] = o.SortedRanges(int(self.ix.seq_len(qid)), 0)
            add = True
            m = params.min_distance_between_ranges
            if m > 0:
                lo, hi = min(qs, qe), max(qs, qe)
                rg = sr.ranges()
                idx = 0
                while idx < len(rg) and rg[idx][0] < lo:
                    idx += 1
                if idx > 0 and abs(lo - rg[idx - 1][1]) < m:
                    add = False
                if add and idx < len(rg) and abs(rg[idx][0] - hi) < m:
                    add = False
            if add:
                for a, b in sr.insert(qs, qe):
                    if abs(b - a) >= params.min_transitive_len:
                        nxt.setdefault(q, []).append((qid, a, b))
        out = []
        for q in sorted(nxt):
            rs = sorted(nxt[q], key=lambda x: (x[0], x[1]))
            merged = [list(rs[0])]
            for r in rs[1:]:
                if merged[-1][0] == r[0] and merged[-1][2] >= r[1]:
                    merged[-1][2] = max(merged[-1][2], r[2])
                else:
                    merged.append(list(r))
            out += [(t, a, b, q) for t, a, b in merged]
        return torch.tensor(out, dtype=torch.int32).view(-1, 4)
