"""Multi-GPU execution: one process per GPU, the alignment index sharded by
target sequence (target_id % world == rank), queries resident on their HOME
rank (the rank that submitted them), the frontier exchanged at every hop of the
transitive closure (src/impg.rs:2376-2594) with torch.distributed (backend
"nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests).

Per hop, in lock step on every rank:
  1. home  : bucket its frontier by owner rank; all_gather of the bucket sizes,
             all_to_all_single of the 16-byte records          (exchange #1)
  2. owner : lookup + CIGAR projection of the records it received, on its shard
  3. owner : hits go back to the record's home with all_to_all_single (#2);
             skipped on the last level in counting mode
  4. home  : hits re-ordered by frontier index (stable: hits of one record
             arrive contiguous and in visit order), visited-set update, next
             frontier (impg.rs:2471-2584)
A frontier record lives on exactly one owner, so the reference's processing
order (frontier order x visit order) is reproduced whatever the arrival order.

`backend` is the per-rank engine: GpuBackend (stage API of libimpg_gpu.so) or,
in CPU tests, a stand-in with the same four methods.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .index import GpuImpg

FR_COLS = 4   # impg_gpu_frontier_t as int32[4]: target_id, start, end, qidx
HIT_COLS = 8  # impg_gpu_hit_t as int32[8]: fidx, query_id, q_first, q_last, t_first, t_last, order, pad
HIT16_COLS = 4  # impg_gpu_hit16_t: the first four of them -- all the visited-set update reads


class GpuBackend:
    """Stage API of the C ABI on torch CUDA tensors (device memory plumbing only)."""

    def __init__(self, index, device):
        self.index = index
        self.device = torch.device("cuda", device)
        self.slice_records = 1 << 26   # frontier records per stage call
        self.pair_budget = 1 << 30     # candidate pairs per projection launch (36 B of slots each)

    def _sync(self):
        # the engine runs on its own non-blocking HIP stream: inputs produced by
        # torch / RCCL on torch's stream must be complete before it reads them
        # (the stage calls synchronise their own stream before returning)
        torch.cuda.current_stream(self.device).synchronize()

    def begin(self, ranges_t, n, params):
        self._sync()
        fr = torch.empty((max(n, 1), FR_COLS), dtype=torch.int32, device=self.device)
        self_iv = torch.empty((max(n, 1), FR_COLS), dtype=torch.int32, device=self.device)
        nf = self.index.stage_begin(ranges_t.data_ptr(), n, params, fr.data_ptr(), self_iv.data_ptr())
        return fr[:nf], self_iv[:n]

    def route(self, frontier, world):
        """Stable partition by owner rank (target_id % world): (records grouped by owner with qidx :=
        their index in `frontier`, counts per owner).  Native: one 1..10-bit radix sort."""
        self._sync()
        n = frontier.shape[0]
        out = torch.empty((max(n, 1), FR_COLS), dtype=torch.int32, device=self.device)
        counts = self.index.stage_route(frontier.data_ptr() if n else None, n, world, out.data_ptr())
        return out[:n], [int(c) for c in counts]

    def reorder(self, hits, n_frontier):
        """Hits that came home grouped by owner (each block ascending in fidx) -> ascending fidx, stable.
        Native: a counting pass over the frontier indices (impg_gpu_stage_reorder), not a sort."""
        self._sync()
        n = hits.shape[0]
        if n == 0:
            return hits
        hits = hits.contiguous()
        out = torch.empty_like(hits)
        self.index.stage_reorder(hits.data_ptr(), n, hits.shape[1], n_frontier, out.data_ptr())
        return out

    def expand(self, frontier, transitive, params, want_hits=True, compact=False):
        """-> (hits int32[k,8] with fidx indexing `frontier`, accepted count).  Slots whose projection
        returned None (query_id == -1) stay in the list: they are rare and every consumer skips them.
        compact: int32[k,4] records (no target columns): enough for the visited-set update, half the bytes."""
        cols = HIT16_COLS if compact else HIT_COLS
        self._sync()
        n = frontier.shape[0]
        outs, accepted, base = [], 0, 0
        step = self.slice_records
        while base < n:
            m = min(step, n - base)
            sub = frontier[base:base + m]
            counts = torch.empty(m, dtype=torch.int32, device=self.device)
            total = self.index.stage_count(sub.data_ptr(), m, transitive, counts.data_ptr())
            if total > self.pair_budget and m > 1:  # keep one projection launch under the pair budget
                step = max(1, m // 2)
                continue
            if want_hits:
                hits = torch.empty((max(total, 1), cols), dtype=torch.int32, device=self.device)
                accepted += self.index.stage_project(sub.data_ptr(), m, transitive, params, hits.data_ptr(), total, compact=compact)
                if total:
                    h = hits[:total]
                    if base:
                        h[:, 0] += base
                    outs.append(h)
            else:  # counting only: the hits stay in the engine's slot arrays
                accepted += self.index.stage_project(sub.data_ptr(), m, transitive, params, None, total)
            base += m
        if not want_hits or not outs:
            return torch.empty((0, cols), dtype=torch.int32, device=self.device), accepted
        return (outs[0] if len(outs) == 1 else torch.cat(outs)), accepted

    def update(self, frontier, hits, params):
        self._sync()
        nn = self.index.stage_update(frontier.data_ptr() if frontier.shape[0] else None, frontier.shape[0],
                                     hits.data_ptr() if hits.shape[0] else None, hits.shape[0], params,
                                     compact=hits.shape[1] == HIT16_COLS)
        out = torch.empty((max(nn, 1), FR_COLS), dtype=torch.int32, device=self.device)
        self.index.stage_next_frontier(out.data_ptr(), nn)
        return out[:nn]


class ShardStats:
    def __init__(self):
        self.projected = 0        # accepted projections computed on THIS rank's shard
        self.frontier_ranges = 0
        self.levels = 0
        self.ms_lookup = self.ms_project = self.ms_update = self.ms_total = 0.0
        self.pairs = 0
        self.project_launches = 0


class ShardedImpg:
    def __init__(self, backend, rank, world, device=None, chunk_ranges=8192):
        self.backend = backend
        self.rank, self.world = rank, world
        self.device = device if device is not None else torch.device("cpu")
        self.chunk_ranges = chunk_ranges
        self.local = getattr(backend, "index", None)
        self.always_reorder = False  # tests: run the home-side reorder even with one rank
        # collectives run on the compute device (RCCL) unless the process group is
        # gloo, which moves host memory: then tensors hop through the CPU
        self.comm_device = self.device if dist.get_backend() != "gloo" else torch.device("cpu")

    @classmethod
    def from_paf(cls, paths, rank, world, device=0, **kw):
        index = GpuImpg.from_paf(paths, device=device, shard=rank, n_shards=world, **kw)
        return cls(GpuBackend(index, device), rank, world, torch.device("cuda", device))

    # ---- collectives -------------------------------------------------------------
    # bytes one rank sends or receives in ONE all_to_all_single call.  Larger exchanges are cut into
    # rounds: on this stack (RCCL 2.26 / ROCm 7.0) a message past 1 GiB came back with half of its rows
    # wrong (scripts/dbg_a2a.py), and bounded rounds also bound the staging memory.
    A2A_ROUND_BYTES = 512 << 20

    def _all_to_all_rows(self, rows, send_counts):
        """rows grouped by destination rank (send_counts[d] rows each) -> rows
        received, grouped by source rank, and the per-source counts."""
        W = self.world
        sc = torch.as_tensor(send_counts, dtype=torch.int64, device=self.comm_device)
        gathered = [torch.empty_like(sc) for _ in range(W)]
        dist.all_gather(gathered, sc)  # every rank learns the full W x W count matrix
        mat = torch.stack(gathered).cpu()
        recv_counts = mat[:, self.rank].tolist()
        cols = rows.shape[1]
        out = torch.empty((int(sum(recv_counts)), cols), dtype=rows.dtype, device=self.comm_device)
        rows = rows.contiguous().to(self.comm_device)
        # every rank derives the same number of rounds from the same matrix
        busiest = int(max(mat.sum(dim=1).max(), mat.sum(dim=0).max()))
        K = max(1, -(-busiest * cols * rows.element_size() // self.A2A_ROUND_BYTES))
        if K == 1:
            dist.all_to_all_single(out.view(-1), rows.view(-1),
                                   output_split_sizes=[c * cols for c in recv_counts],
                                   input_split_sizes=[int(c) * cols for c in send_counts])
            return out.to(self.device), recv_counts
        soff = np.concatenate([[0], np.cumsum(send_counts)]).astype(np.int64)
        roff = np.concatenate([[0], np.cumsum(recv_counts)]).astype(np.int64)
        cut = lambda c, k: (int(c) * k) // K  # round k moves rows [cut(c,k), cut(c,k+1)) of every (source, destination) block
        for k in range(K):
            ins = [(int(soff[d]) + cut(send_counts[d], k), int(soff[d]) + cut(send_counts[d], k + 1)) for d in range(W)]
            outs = [(int(roff[s]) + cut(recv_counts[s], k), int(roff[s]) + cut(recv_counts[s], k + 1)) for s in range(W)]
            if W == 1:  # one block each way: the slices are already contiguous
                src, dst = rows[ins[0][0]:ins[0][1]], out[outs[0][0]:outs[0][1]]
            else:
                src = torch.cat([rows[a:b] for a, b in ins])
                dst = torch.empty((sum(b - a for a, b in outs), cols), dtype=rows.dtype, device=self.comm_device)
            dist.all_to_all_single(dst.view(-1), src.view(-1),
                                   output_split_sizes=[(b - a) * cols for a, b in outs],
                                   input_split_sizes=[(b - a) * cols for a, b in ins])
            if W > 1:
                pos = 0
                for a, b in outs:
                    out[a:b] = dst[pos:pos + (b - a)]
                    pos += b - a
        return out.to(self.device), recv_counts

    def _any(self, flag):
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self.comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t.item())

    def _route(self, front):
        if hasattr(self.backend, "route"):
            return self.backend.route(front, self.world)
        W = self.world  # generic (CPU stand-in backends)
        owner = torch.remainder(front[:, 0].to(torch.int64) & 0xFFFFFFFF, W)
        order = torch.argsort(owner, stable=True)
        send = front[order].clone()
        send[:, 3] = order.to(torch.int32)
        return send, torch.bincount(owner, minlength=W).tolist()

    # ---- one hop --------------------------------------------------------------------
    def _hop(self, front, transitive, params, need_hits, compact=False):
        W = self.world
        send, send_counts = self._route(front)  # grouped by owner; column 3 = the home frontier index, echoed back
        recv, recv_counts = self._all_to_all_rows(send, send_counts)
        hits, accepted = self.backend.expand(recv, transitive, params, want_hits=need_hits, compact=compact)
        if not need_hits:
            return None, accepted, recv.shape[0]
        # hits are ordered by fidx and `recv` is grouped by source rank, so hits are
        # already grouped by the rank they must return to
        bounds = torch.as_tensor(np.cumsum([0] + recv_counts), dtype=torch.int64, device=self.device)
        fidx = hits[:, 0].to(torch.int64)
        back_counts = (torch.searchsorted(fidx, bounds[1:], right=False) - torch.searchsorted(fidx, bounds[:-1], right=False)).tolist()
        hits[:, 0] = recv[fidx, 3]  # (the hit list is ours: relabel in place)
        back, _ = self._all_to_all_rows(hits, back_counts)
        if W == 1 and not self.always_reorder:
            return back, accepted, recv.shape[0]  # one source: already in home frontier order
        if hasattr(self.backend, "reorder"):
            return self.backend.reorder(back, front.shape[0]), accepted, recv.shape[0]
        perm = torch.argsort(back[:, 0].to(torch.int64), stable=True)  # generic (CPU stand-in backends)
        return back[perm].contiguous(), accepted, recv.shape[0]

    # ---- batches ----------------------------------------------------------------------
    def _run_chunk(self, ranges_t, n, params, collect):
        """One chunk of this rank's queries, in lock step with the other ranks.
        collect: None, or a list that receives (level, frontier, hits) at HOME."""
        transitive = bool(params.transitive)
        st = ShardStats()
        if transitive:
            front, self_iv = self.backend.begin(ranges_t, n, params)
        else:
            r = ranges_t.view(torch.int32).view(-1, 3)[:n]
            front = torch.cat([r, torch.arange(n, dtype=torch.int32, device=r.device).view(-1, 1)], dim=1).contiguous()
            self_iv = front
        depth = 0
        while True:
            more = self._any(front.shape[0] > 0)
            if not more or (transitive and params.max_depth > 0 and depth >= params.max_depth):
                break
            last = (not transitive) or (params.max_depth > 0 and depth + 1 >= params.max_depth)
            need_hits = (collect is not None) or not last
            # nobody reads result rows at home in a counting run: the owners send back 16-byte records
            hits, accepted, n_looked = self._hop(front, transitive, params, need_hits, compact=collect is None)
            st.projected += accepted
            st.frontier_ranges += n_looked
            st.levels += 1
            if collect is not None:
                collect.append((depth, front, hits))
            if last:
                break
            front = self.backend.update(front, hits, params)
            depth += 1
        return st, self_iv

    def _chunks(self, n):
        n_max = torch.tensor([n], dtype=torch.int64, device=self.comm_device)
        dist.all_reduce(n_max, op=dist.ReduceOp.MAX)
        n_chunks = max(1, -(-int(n_max.item()) // self.chunk_ranges))
        for c in range(n_chunks):
            b = min(n, c * self.chunk_ranges)
            e = min(n, b + self.chunk_ranges)
            yield b, e

    def query_batch_stats(self, ranges_t, n, params):
        """Counting mode (bench): results stay where they are computed."""
        tot = ShardStats()
        item = _lib.RANGE_DTYPE.itemsize
        if self.local is not None:
            self.local.stage_timing(reset=True)
        for b, e in self._chunks(n):
            st, _ = self._run_chunk(ranges_t[b * item:], e - b, params, None)
            tot.projected += st.projected
            tot.frontier_ranges += st.frontier_ranges
            tot.levels = max(tot.levels, st.levels)
        if self.local is not None:
            ms, launches = self.local.stage_timing(reset=True)
            tot.ms_lookup, tot.ms_project, tot.ms_update = ms
            tot.ms_total = sum(ms)
            tot.project_launches = launches
        return tot

    def query_batch(self, ranges, params):
        """Full results for this rank's queries, in the reference's emission order:
        list (per range) of numpy INTERVAL_DTYPE arrays."""
        ranges = np.ascontiguousarray(ranges, dtype=_lib.RANGE_DTYPE)
        n = ranges.size
        ranges_t = torch.from_numpy(ranges.view(np.uint8).copy()).to(self.device)
        item = _lib.RANGE_DTYPE.itemsize
        transitive = bool(params.transitive)
        out = [[] for _ in range(n)]
        for b, e in self._chunks(n):
            collect = []
            _, self_iv = self._run_chunk(ranges_t[b * item:], e - b, params, collect)
            sv = self_iv.cpu().numpy()
            for q in range(e - b):
                t, s, en = int(sv[q, 0]) & 0xFFFFFFFF, int(sv[q, 1]), int(sv[q, 2])
                if (not transitive) or s < en:
                    out[b + q].append((t, s, en, t, s, en))
            for depth, front, hits in collect:
                f = front.cpu().numpy()
                h = hits.cpu().numpy()
                for k in range(h.shape[0]):
                    if int(h[k, 1]) == -1:
                        continue  # projection returned None
                    fi = int(h[k, 0])
                    qs, qe = int(h[k, 2]), int(h[k, 3])
                    if transitive and params.min_output_length >= 0 and abs(qe - qs) < params.min_output_length:
                        continue
                    out[b + int(f[fi, 3])].append((int(h[k, 1]) & 0xFFFFFFFF, qs, qe, int(f[fi, 0]) & 0xFFFFFFFF,
                                                   int(h[k, 4]), int(h[k, 5])))
        return [np.array(x, dtype=_lib.INTERVAL_DTYPE) for x in out]
