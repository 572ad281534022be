"""Host-side mirror of the reference's index interface (trait ImpgIndex,
src/impg_index.rs:21-121) over the C ABI.  Method names and argument meaning
follow the trait; the batch_* methods are the fast path (one launch sequence for
many ranges)."""
import os
import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import IMPG_E_UNSUPPORTED, INTERVAL_DTYPE, RANGE_DTYPE, RECORD_DTYPE, ImpgGpuError, Params, Stats, check, lib


def make_params(transitive=False, dfs=False, max_depth=2, min_transitive_len=101, min_distance_between_ranges=10,
                min_output_length=None, min_identity=None, store_cigar=False, multi_impg=False,
                original_sequence_coordinates=False, consider_strandness=False):
    """CLI defaults of src/main.rs:4259-4285."""
    return Params(int(transitive), int(dfs), max_depth, min_transitive_len, min_distance_between_ranges,
                  -1 if min_output_length is None else min_output_length,
                  math.nan if min_identity is None else float(min_identity), int(store_cigar), int(multi_impg),
                  int(original_sequence_coordinates), int(consider_strandness))


class QueryResults:
    """Vec<AdjustedInterval> per range, in the reference's emission order."""

    def __init__(self, handle, owner, copy=True):
        """copy=False: the arrays are views of the library's buffers (valid while this object lives); a
        transitive batch is hundreds of millions of rows, not worth duplicating."""
        self._h = handle
        self._owner = owner
        L = lib()
        self.n_ranges = L.impg_gpu_results_num_ranges(handle)
        total = L.impg_gpu_results_total(handle)
        self.total = total
        self.offsets = np.ctypeslib.as_array(C.cast(L.impg_gpu_results_offsets(handle), C.POINTER(C.c_uint64)),
                                             shape=(self.n_ranges + 1,)).copy()
        if total:
            raw = np.ctypeslib.as_array(C.cast(L.impg_gpu_results_intervals(handle), C.POINTER(C.c_uint8)),
                                        shape=(total * INTERVAL_DTYPE.itemsize,))
            self.intervals = raw.view(INTERVAL_DTYPE).copy() if copy else raw.view(INTERVAL_DTYPE)
        else:
            self.intervals = np.zeros(0, dtype=INTERVAL_DTYPE)
        self.projected = L.impg_gpu_results_projected(handle)
        self.cigar_off = self.cigar_ops = None
        cop = L.impg_gpu_results_cigar_offsets(handle)
        if cop:
            self.cigar_off = np.ctypeslib.as_array(C.cast(cop, C.POINTER(C.c_uint64)), shape=(total + 1,)).copy()
            nops = int(self.cigar_off[-1])
            self.cigar_ops = (np.ctypeslib.as_array(C.cast(L.impg_gpu_results_cigar_ops(handle), C.POINTER(C.c_uint32)),
                                                    shape=(nops,)).copy() if nops else np.zeros(0, dtype=np.uint32))

    def __getitem__(self, i):
        return self.intervals[self.offsets[i]:self.offsets[i + 1]]

    def timing(self):
        """(seconds in the engine, seconds copying back + assembling) of the call (impg_gpu_results_timing)."""
        a, b = C.c_double(0), C.c_double(0)
        lib().impg_gpu_results_timing(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def cigars(self, i):
        """Vec<CigarOp> (packed u32 ops) of every interval of range i; needs store_cigar."""
        a, b = int(self.offsets[i]), int(self.offsets[i + 1])
        return [self.cigar_ops[int(self.cigar_off[k]):int(self.cigar_off[k + 1])] for k in range(a, b)]

    def __len__(self):
        return self.n_ranges

    def bed(self, range_names=None, merge_distance=0, params=None):
        """output_results_bed over every range (main.rs:11849-11892)."""
        L = lib()
        p = params or make_params()
        arr = None
        if range_names is not None:
            arr = (C.c_char_p * len(range_names))(*[None if s is None else s.encode() for s in range_names])
        text = C.c_void_p(None)
        ln = C.c_size_t(0)
        check(L.impg_gpu_results_bed(self._h, self._owner._h, arr, C.byref(p), merge_distance, C.byref(text), C.byref(ln)))
        try:
            return C.string_at(text, ln.value).decode()
        finally:
            _lib.free(text)

    def paf(self, range_names=None, merge_distance=0, params=None, fmt="paf"):
        """output_results_paf / output_results_bedpe over every range (main.rs:11894-12103); the batch
        must have been queried with store_cigar = True."""
        L = lib()
        p = params or make_params(store_cigar=True)
        arr = None
        if range_names is not None:
            arr = (C.c_char_p * len(range_names))(*[None if s is None else s.encode() for s in range_names])
        text = C.c_void_p(None)
        ln = C.c_size_t(0)
        check(L.impg_gpu_results_paf(self._h, self._owner._h, arr, C.byref(p), merge_distance, {"paf": 0, "bedpe": 1}[fmt],
                                     C.byref(text), C.byref(ln)))
        try:
            return C.string_at(text, ln.value).decode()
        finally:
            _lib.free(text)

    def __del__(self):
        if getattr(self, "_h", None) and not getattr(self, "_borrowed", False):
            try:
                lib().impg_gpu_results_free(self._h)
            except Exception:
                pass
            self._h = None


_HIP = None


def _hip_memcpy_d2h(dst, src, nbytes):
    """hipMemcpy device -> host through the HIP runtime the library itself links (tests and probes read device rows back)."""
    global _HIP
    if _HIP is None:
        _HIP = C.CDLL("libamdhip64.so")
        _HIP.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _HIP.hipMemcpy.restype = C.c_int
    if nbytes:
        rc = _HIP.hipMemcpy(dst, src, nbytes, 2)
        if rc != 0:
            raise ImpgGpuError(_lib.IMPG_E_HIP, "hipMemcpy (device to host) failed: %d" % rc)


class DeviceRows:
    """impg_gpu_device_rows_t: the rows of a batch left in HBM (impg_gpu_query_batch_device).  Holds one of the index's
    engines until freed (free() or garbage collection)."""

    def __init__(self, handle, owner, n):
        self._h = handle
        self._owner = owner  # keeps the index alive
        self._n = int(n)
        st = Stats()
        lib().impg_gpu_device_rows_stats(self._h, C.byref(st))
        self.stats = st
        self.projected = int(st.projected)
        self.place_ms = float(lib().impg_gpu_device_rows_place_ms(self._h))

    def parts(self):
        out = []
        for k in range(lib().impg_gpu_device_rows_num_parts(self._h)):
            d = _lib.DevicePart()
            check(lib().impg_gpu_device_rows_part(self._h, k, C.byref(d)))
            out.append(d)
        return out

    def part_to_host(self, k):
        """One part copied back: (first_range, level, query_id[n], coords[n, 4], source[n], frontier[n_frontier])."""
        d = self.parts()[k]
        n, nf, st = int(d.n_slots), int(d.n_frontier), int(d.slot_stride)
        co = np.empty((n, 4), dtype=np.int32)
        fr = np.empty(nf, dtype=_lib.FRONTIER_DTYPE)
        if st == 1:
            qid = np.empty(n, dtype=np.uint32)
            src = np.empty(n, dtype=np.uint32)
            _hip_memcpy_d2h(qid.ctypes.data, d.query_id, n * 4)
            _hip_memcpy_d2h(src.ctypes.data, d.source, n * 4)
        else:  # query_id and source side by side (source = query_id + 1 word)
            assert d.source == d.query_id + 4 and st == 2
            both = np.empty((n, 2), dtype=np.uint32)
            _hip_memcpy_d2h(both.ctypes.data, d.query_id, n * 8)
            qid, src = np.ascontiguousarray(both[:, 0]), np.ascontiguousarray(both[:, 1])
        _hip_memcpy_d2h(co.ctypes.data, d.coords, n * 16)
        _hip_memcpy_d2h(fr.ctypes.data, d.frontier, nf * 16)
        return int(d.first_range), int(d.level), qid, co, src, fr

    def ordered_to_host(self, k=0):
        """IMPG_ROWS_ORDERED part k copied back: (first_range, rows[INTERVAL_DTYPE], offsets[n_ranges + 1])."""
        d = self.parts()[k]
        rows = np.empty(int(d.n_slots), dtype=INTERVAL_DTYPE)
        off = np.empty(int(d.n_ranges) + 1, dtype=np.uint32)
        _hip_memcpy_d2h(rows.ctypes.data, d.rows, rows.nbytes)
        _hip_memcpy_d2h(off.ctypes.data, d.offsets, off.nbytes)
        return int(d.first_range), rows, off

    def check(self, counts=True, checksums=True):
        """impg_gpu_device_rows_check: per-range counts / checksums recomputed from the rows in HBM."""
        n = self._n
        cnt = np.zeros(n, dtype=np.uint64) if counts else None
        ck = np.zeros(n, dtype=np.uint64) if checksums else None
        check(lib().impg_gpu_device_rows_check(self._h, cnt.ctypes.data if counts else None, ck.ctypes.data if checksums else None))
        return cnt, ck

    def free(self):
        if self._h:
            lib().impg_gpu_device_rows_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PreparedMask:
    """A masked_regions map already in the C layout (impg_gpu_mask_t + the arrays it points into): per-call users --
    the shape of partition.rs:359-391 -- convert once per change of the map, not once per call."""

    def __init__(self, m, arrays):
        self.m, self.arrays = m, arrays


def prepare_mask(masked_regions):
    return PreparedMask(*GpuImpg._mask(masked_regions))


class GpuImpg:
    """`impl ImpgIndex` backed by the HIP engine."""

    def __init__(self, handle):
        self._h = handle

    # ---- construction (Impg::from_multi_alignment_records / load) ------------
    @classmethod
    def from_records(cls, records, ops, seq_len, bidirectional=True, order=_lib.ORDER_COITREES, device=0,
                     file_first=None, devices=None, lanes=2, comm=None):
        """file_first: first record of every alignment file (records_by_file of the reference); only the
        MultiImpg tie order observes it.
        devices=[...]: ONE handle over several GPUs of this process (impg_gpu_index_create_multi).
        comm=Comm: this rank's shard of an index sharded one process per GPU (impg_gpu_index_create_rank);
        queries on it are collective calls."""
        rec = np.ascontiguousarray(records, dtype=RECORD_DTYPE)
        ops = np.ascontiguousarray(ops, dtype=np.uint32)
        sl = np.ascontiguousarray(seq_len, dtype=np.int64)
        h = C.c_void_p(None)
        ff = None if file_first is None else np.ascontiguousarray(file_first, dtype=np.uint64)
        ffp, ffn = (None, 0) if ff is None else (ff.ctypes.data, ff.size)
        if devices is not None:
            dv = np.ascontiguousarray(devices, dtype=np.int32)
            check(lib().impg_gpu_index_create_multi(rec.ctypes.data, rec.size, ops.ctypes.data, ops.size, sl.ctypes.data, sl.size,
                                                    ffp, ffn, int(bidirectional), order, dv.ctypes.data, dv.size, lanes, C.byref(h)))
        elif comm is not None:
            check(lib().impg_gpu_index_create_rank(rec.ctypes.data, rec.size, ops.ctypes.data, ops.size, sl.ctypes.data, sl.size,
                                                   ffp, ffn, int(bidirectional), order, device, comm._h, C.byref(h)))
        elif ff is not None:
            check(lib().impg_gpu_index_create_files(rec.ctypes.data, rec.size, ops.ctypes.data, ops.size, sl.ctypes.data, sl.size,
                                                    ffp, ffn, int(bidirectional), order, device, C.byref(h)))
        else:
            check(lib().impg_gpu_index_create(rec.ctypes.data, rec.size, ops.ctypes.data, ops.size, sl.ctypes.data, sl.size,
                                              int(bidirectional), order, device, C.byref(h)))
        ix = cls(h)
        ix._comm = comm  # the communicator outlives the index
        return ix

    @classmethod
    def load_impg(cls, impg_path, alignment_files, order=_lib.ORDER_COITREES, device=0):
        """impg_gpu_index_load_impg: the reference's own IMPGIDX2 index file + the alignment files it was built from."""
        if isinstance(alignment_files, str):
            alignment_files = [alignment_files]
        arr = (C.c_char_p * len(alignment_files))(*[p.encode() for p in alignment_files])
        h = C.c_void_p(None)
        check(lib().impg_gpu_index_load_impg(os.fsencode(impg_path), arr, len(alignment_files), order, device, C.byref(h)))
        return cls(h)

    @classmethod
    def from_tracepoints(cls, records, tracepoints, seq_len, query_deltas=None, diffs=None, fastga=False, trace_spacing=0,
                         max_complexity=0, bidirectional=True, order=_lib.ORDER_COITREES, device=0, devices=None, lanes=2):
        """impg_gpu_index_create_tracepoints: an index over tracepoint alignments (.1aln / .tpa content handed over as
        arrays); every query on it runs in approximate mode (approximate_mode = true of the trait)."""
        rec = np.ascontiguousarray(records, dtype=_lib.TP_RECORD_DTYPE)
        tp = np.ascontiguousarray(tracepoints, dtype=np.int32)
        qd = None if query_deltas is None else np.ascontiguousarray(query_deltas, dtype=np.int32)
        df = None if diffs is None else np.ascontiguousarray(diffs, dtype=np.int32)
        sl = np.ascontiguousarray(seq_len, dtype=np.int64)
        mode = _lib.TpMode(int(bool(fastga)), int(trace_spacing), int(max_complexity))
        h = C.c_void_p(None)
        if devices is not None:  # sharded by target sequence over the GPUs of this process
            dv = np.ascontiguousarray(devices, dtype=np.int32)
            check(lib().impg_gpu_index_create_tracepoints_multi(rec.ctypes.data, rec.size, tp.ctypes.data, None if qd is None else qd.ctypes.data,
                                                                None if df is None else df.ctypes.data, tp.size, C.byref(mode), sl.ctypes.data,
                                                                sl.size, int(bidirectional), order, dv.ctypes.data, dv.size, lanes, C.byref(h)))
            return cls(h)
        check(lib().impg_gpu_index_create_tracepoints(rec.ctypes.data, rec.size, tp.ctypes.data, None if qd is None else qd.ctypes.data,
                                                      None if df is None else df.ctypes.data, tp.size, C.byref(mode), sl.ctypes.data,
                                                      sl.size, int(bidirectional), order, device, C.byref(h)))
        return cls(h)

    def save(self, path):
        """impg_gpu_index_save: the built index (device arrays + sequence table) as one file."""
        check(lib().impg_gpu_index_save(self._h, os.fsencode(path)))

    @classmethod
    def load(cls, path, device=0, devices=None, lanes=2, comm=None):
        """impg_gpu_index_load: what `save` wrote, back in HBM without touching the alignment files.
        devices=[...]: a saved multi handle (impg_gpu_index_load_multi); comm=Comm: this rank's saved shard
        (impg_gpu_index_load_rank)."""
        h = C.c_void_p(None)
        if devices is not None:
            dv = np.ascontiguousarray(devices, dtype=np.int32)
            check(lib().impg_gpu_index_load_multi(os.fsencode(path), dv.ctypes.data, dv.size, lanes, C.byref(h)))
        elif comm is not None:
            check(lib().impg_gpu_index_load_rank(os.fsencode(path), device, comm._h, C.byref(h)))
        else:
            check(lib().impg_gpu_index_load(os.fsencode(path), device, C.byref(h)))
        ix = cls(h)
        ix._comm = comm
        return ix

    @classmethod
    def from_paf(cls, paths, bidirectional=True, order=_lib.ORDER_COITREES, device=0, devices=None, lanes=2, comm=None):
        """devices / comm: as in from_records (impg_gpu_index_create_from_paf_multi / _rank)."""
        if isinstance(paths, str):
            paths = [paths]
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        h = C.c_void_p(None)
        if devices is not None:
            dv = np.ascontiguousarray(devices, dtype=np.int32)
            check(lib().impg_gpu_index_create_from_paf_multi(arr, len(paths), int(bidirectional), order, dv.ctypes.data, dv.size,
                                                             lanes, C.byref(h)))
        elif comm is not None:
            check(lib().impg_gpu_index_create_from_paf_rank(arr, len(paths), int(bidirectional), order, device, comm._h, C.byref(h)))
        else:
            check(lib().impg_gpu_index_create_from_paf(arr, len(paths), int(bidirectional), order, device, C.byref(h)))
        ix = cls(h)
        ix._comm = comm
        return ix

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().impg_gpu_index_destroy(self._h)
            except Exception:  # interpreter shutdown: the binding module may already be torn down
                pass
            self._h = None

    # ---- seq_index() / target_ids() / num_targets() ---------------------------
    def num_seqs(self):
        return lib().impg_gpu_num_seqs(self._h)

    def seq_name(self, i):
        s = lib().impg_gpu_seq_name(self._h, i)
        return None if s is None else s.decode()

    def seq_len(self, i):
        return lib().impg_gpu_seq_len(self._h, i)

    def seq_id(self, name):
        r = lib().impg_gpu_seq_id(self._h, name.encode())
        return None if r < 0 else int(r)

    def num_targets(self):
        return lib().impg_gpu_num_targets(self._h)

    def target_ids(self):
        n = lib().impg_gpu_target_ids(self._h, None, 0)
        out = np.zeros(n, dtype=np.uint32)
        lib().impg_gpu_target_ids(self._h, out.ctypes.data, n)
        return out

    def num_entries(self):
        return lib().impg_gpu_num_entries(self._h)

    def num_records(self):
        return lib().impg_gpu_num_records(self._h)

    def device_bytes(self):
        return lib().impg_gpu_device_bytes(self._h)

    def set_option(self, key, value):
        check(lib().impg_gpu_set_option(self._h, key.encode(), int(value)))

    def counter(self, key):
        """impg_gpu_get_counter: "walk_launches", "walk_fallbacks", "walk_members"."""
        v = C.c_int64(0)
        check(lib().impg_gpu_get_counter(self._h, key.encode(), C.byref(v)))
        return v.value

    # ---- queries ---------------------------------------------------------------
    @staticmethod
    def _ranges(ranges):
        if isinstance(ranges, np.ndarray) and ranges.dtype == RANGE_DTYPE:
            return np.ascontiguousarray(ranges)
        a = np.zeros(len(ranges), dtype=RANGE_DTYPE)
        for i, (t, s, e) in enumerate(ranges):
            a[i] = (t, s, e)
        return a

    @staticmethod
    def _mask(masked_regions):
        """{sequence id: (sequence_length, [(start, end), ...])} -> (impg_gpu_mask_t, arrays kept alive).  A
        PreparedMask (prepare_mask: the conversion done once for many calls) passes through."""
        if isinstance(masked_regions, PreparedMask):
            return masked_regions.m, masked_regions.arrays
        ids = sorted(masked_regions)
        seq = np.array(ids, dtype=np.uint32)
        slen = np.array([masked_regions[i][0] for i in ids], dtype=np.int32)
        off = np.zeros(len(ids) + 1, dtype=np.uint64)
        flat = []
        for k, i in enumerate(ids):
            flat.extend(masked_regions[i][1])
            off[k + 1] = len(flat)
        rng = np.ascontiguousarray(np.array(flat, dtype=np.int32).reshape(-1, 2))
        m = _lib.Mask(len(ids), seq.ctypes.data, slen.ctypes.data, off.ctypes.data, rng.ctypes.data)
        return m, (seq, slen, off, rng)

    def query_batch(self, ranges, params=None, masked_regions=None, subset_keep=None, copy=True, **kw):
        """masked_regions: one map for the whole batch (impg_gpu_query_batch_masked); every range starts
        from its own clone of it, as the reference's per-range calls do (impg.rs:2077-2081).
        subset_keep: uint8[num_seqs], the host's SubsetFilter::matches verdict per sequence id."""
        p = params or make_params(**kw)
        r = self._ranges(ranges)
        h = C.c_void_p(None)
        if subset_keep is not None:
            keep = np.ascontiguousarray(subset_keep, dtype=np.uint8)
            if keep.size != self.num_seqs():
                raise _lib.ImpgGpuError(_lib.IMPG_E_INVALID, "subset_keep needs one entry per sequence")
            m, keepalive = self._mask(masked_regions) if masked_regions is not None else (None, None)
            check(lib().impg_gpu_query_batch_filtered(self._h, r.ctypes.data, r.size, C.byref(p),
                                                      C.byref(m) if m is not None else None, keep.ctypes.data, C.byref(h)))
            del keepalive
        elif masked_regions is not None:
            m, keep = self._mask(masked_regions)
            check(lib().impg_gpu_query_batch_masked(self._h, r.ctypes.data, r.size, C.byref(p), C.byref(m), C.byref(h)))
            del keep
        else:
            check(lib().impg_gpu_query_batch(self._h, r.ctypes.data, r.size, C.byref(p), C.byref(h)))
        return QueryResults(h, self, copy=copy)

    def query_batch_stream(self, ranges, consumer, params=None, masked_regions=None, subset_keep=None, chunk_ranges=0,
                           max_block_bytes=0, copy=False, **kw):
        """impg_gpu_query_batch_stream: the rows of a batch too big for one result object, chunk by chunk in range order.
        consumer(first_range, QueryResults) is called for every chunk (copy=False: its arrays are views of the library's
        pinned block, valid only during the call); a truthy return stops the stream.  Returns the projections counted."""
        p = params or make_params(**kw)
        r = self._ranges(ranges)
        keep = None
        if subset_keep is not None:
            keep = np.ascontiguousarray(subset_keep, dtype=np.uint8)
            if keep.size != self.num_seqs():
                raise _lib.ImpgGpuError(_lib.IMPG_E_INVALID, "subset_keep needs one entry per sequence")
        m, keepalive = self._mask(masked_regions) if masked_regions is not None else (None, None)
        failure = []

        def trampoline(ctx, chunk, first):
            try:
                q = QueryResults.__new__(QueryResults)
                q._borrowed = True
                QueryResults.__init__(q, C.c_void_p(chunk), self, copy=copy)
                return 1 if consumer(int(first), q) else 0
            except BaseException as e:  # (an exception must not unwind through the library's thread)
                failure.append(e)
                return 1

        cb = _lib.STREAM_CB(trampoline)
        proj = C.c_uint64(0)
        rc = lib().impg_gpu_query_batch_stream(self._h, r.ctypes.data, r.size, C.byref(p), C.byref(m) if m is not None else None,
                                               None if keep is None else keep.ctypes.data, chunk_ranges, max_block_bytes,
                                               C.cast(cb, C.c_void_p), None, C.byref(proj))
        del keepalive
        if failure:
            raise failure[0]
        if rc != _lib.IMPG_E_CANCELLED:
            check(rc)
        return proj.value

    def query_batch_bed(self, ranges, params=None, merge_distance=0, range_names=None, subset_keep=None, timing=False, raw=False, **kw):
        """impg_gpu_query_batch_bed: query + both BED merges on the device + text (what `impg query -o bed` prints)."""
        p = params or make_params(**kw)
        r = self._ranges(ranges)
        arr = None
        if range_names is not None:
            arr = (C.c_char_p * len(range_names))(*[None if s is None else s.encode() for s in range_names])
        keep = None if subset_keep is None else np.ascontiguousarray(subset_keep, dtype=np.uint8)
        text = C.c_void_p(None)
        ln = C.c_size_t(0)
        sec = (C.c_double * 3)()
        check(lib().impg_gpu_query_batch_bed(self._h, r.ctypes.data, r.size, C.byref(p), None if keep is None else keep.ctypes.data,
                                             merge_distance, arr, C.byref(text), C.byref(ln), sec))
        try:
            out = C.string_at(text, ln.value)
        finally:
            _lib.free(text)
        out = out if raw else out.decode()
        return (out, list(sec)) if timing else out

    def approximate(self):
        """True for an index built from tracepoints (every answer is the reference's approximate mode)."""
        return bool(lib().impg_gpu_index_approximate(self._h))

    def _check_mode(self, approximate_mode):
        # The reference takes approximate_mode per call (impg.rs:1899): on a CIGAR index it then finds no tracepoints
        # and skips every hit, on a tracepoint index without it it realigns (WFA, not built here).  Answering in the
        # index's own mode instead would hand a caller ported from the trait different rows without a word.
        # None (the default) = the index's own mode; only an explicit mismatch is refused, as the C entry points do
        # through params (impg_gpu_index_approximate, IMPG_E_UNSUPPORTED).
        if approximate_mode is not None and bool(approximate_mode) != self.approximate():
            raise ImpgGpuError(IMPG_E_UNSUPPORTED, "approximate_mode=%s on an index built from %s" %
                               (bool(approximate_mode), "tracepoints" if self.approximate() else "CIGARs"))

    def query(self, target_id, range_start, range_end, store_cigar=False, min_gap_compressed_identity=None,
              sequence_index=None, approximate_mode=None):
        """ImpgIndex::query (impg_index.rs:26-35)."""
        # approximate_mode is a property of the index here: one built from tracepoints (from_tracepoints) answers
        # in approximate mode, one built from CIGARs in exact mode; a call that asks for the other mode is refused
        self._check_mode(approximate_mode)
        p = make_params(transitive=False, store_cigar=store_cigar, min_identity=min_gap_compressed_identity)
        return self.query_batch([(target_id, range_start, range_end)], p)[0]

    def query_transitive_bfs(self, target_id, range_start, range_end, masked_regions=None, max_depth=2,
                             min_transitive_len=101, min_distance_between_ranges=10, min_output_length=None,
                             store_cigar=False, min_gap_compressed_identity=None, sequence_index=None,
                             approximate_mode=None, subset_filter=None):
        """ImpgIndex::query_transitive_bfs (impg_index.rs:79-94)."""
        self._check_mode(approximate_mode)
        p = make_params(True, False, max_depth, min_transitive_len, min_distance_between_ranges, min_output_length,
                        min_gap_compressed_identity, store_cigar)
        return self.query_batch([(target_id, range_start, range_end)], p, masked_regions=masked_regions,
                                subset_keep=subset_filter)[0]

    def query_transitive_dfs(self, target_id, range_start, range_end, masked_regions=None, max_depth=2,
                             min_transitive_len=101, min_distance_between_ranges=10, min_output_length=None,
                             store_cigar=False, min_gap_compressed_identity=None, sequence_index=None,
                             approximate_mode=None, subset_filter=None):
        """ImpgIndex::query_transitive_dfs (impg_index.rs:63-77)."""
        self._check_mode(approximate_mode)
        p = make_params(True, True, max_depth, min_transitive_len, min_distance_between_ranges, min_output_length,
                        min_gap_compressed_identity, store_cigar)
        return self.query_batch([(target_id, range_start, range_end)], p, masked_regions=masked_regions,
                                subset_keep=subset_filter)[0]

    def query_batch_stats(self, ranges, params=None, counts=True, checksums=True, device_ptr=None, n=None, **kw):
        """Throughput form: results stay in HBM; returns (Stats, counts, checksums)."""
        p = params or make_params(**kw)
        st = Stats()
        if device_ptr is None:
            r = self._ranges(ranges)
            n = r.size
        cnt = np.zeros(n, dtype=np.uint64) if counts else None
        ck = np.zeros(n, dtype=np.uint64) if checksums else None
        cp = cnt.ctypes.data if counts else None
        kp = ck.ctypes.data if checksums else None
        if device_ptr is None:
            check(lib().impg_gpu_query_batch_stats(self._h, r.ctypes.data, n, C.byref(p), cp, kp, C.byref(st)))
        else:
            check(lib().impg_gpu_query_batch_stats_dev(self._h, device_ptr, n, C.byref(p), cp, kp, C.byref(st)))
        return st, cnt, ck

    def query_batch_device(self, ranges, params=None, device_ptr=None, n=None, layout=_lib.ROWS_ATTRIBUTED, **kw):
        """impg_gpu_query_batch_device: every result row left in HBM, attributable (DeviceRows).  device_ptr / n: the
        ranges are already on the device."""
        p = params or make_params(**kw)
        h = C.c_void_p(None)
        if device_ptr is None:
            r = self._ranges(ranges)
            n = r.size
            check(lib().impg_gpu_query_batch_device(self._h, r.ctypes.data, r.size, 0, C.byref(p), layout, C.byref(h)))
        else:
            check(lib().impg_gpu_query_batch_device(self._h, device_ptr, n, 1, C.byref(p), layout, C.byref(h)))
        return DeviceRows(h, self, n)

    def hop_profile(self, reset=True):
        """impg_gpu_index_hop_profile: array [shards][8 hops][12 fields] (HOP_PROFILE_FIELDS), zeros for a plain index."""
        n = C.c_size_t(0)
        check(lib().impg_gpu_index_hop_profile(self._h, None, 0, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.float64)
        if n.value:
            check(lib().impg_gpu_index_hop_profile(self._h, out.ctypes.data, out.size, 1 if reset else 0, C.byref(n)))
        return out[:n.value].reshape(-1, 8, 12)

    def shard_info(self):
        """(rank, world, lanes, owner[num_seqs]) of a sharded index; rank -1 = a multi-GPU handle; (0, 1, 1, None) = plain."""
        r, w, l = C.c_int(0), C.c_int(1), C.c_int(1)
        owner = np.zeros(self.num_seqs(), dtype=np.uint32)
        check(lib().impg_gpu_index_shard_info(self._h, C.byref(r), C.byref(w), C.byref(l), owner.ctypes.data, owner.size))
        return r.value, w.value, l.value, (owner if w.value > 1 or r.value != 0 else None)


HOP_PROFILE_FIELDS = ["hops", "route_s", "gather_sizes_s", "records_out_s", "owner_expand_s", "gather_hits_s", "hits_home_s", "reorder_s",
                      "bytes_records_out", "bytes_hits_out", "records_in", "hits_home"]


def shard_assign(entries_per_target, n_shards):
    """impg_gpu_shard_assign: owner shard of every target (greedy bin-packing by entry count); host-only."""
    c = np.ascontiguousarray(entries_per_target, dtype=np.uint64)
    out = np.zeros(c.size, dtype=np.uint32)
    check(lib().impg_gpu_shard_assign(c.ctypes.data, c.size, n_shards, out.ctypes.data))
    return out


class Comm:
    """impg_gpu_comm_t: the transport between the ranks of an index sharded one process per GPU."""

    def __init__(self, handle, rank, world, lanes, keep=None):
        self._h, self.rank, self.world, self.lanes = handle, rank, world, lanes
        self._keep = keep

    @classmethod
    def rccl(cls, rank, world, device, lanes=2, group=None):
        """RCCL communicators (one per lane).  Rank 0 makes the unique ids; torch.distributed (already
        initialised by the launcher) carries them to the other ranks -- plumbing only: every collective of
        a query runs inside libimpg_gpu.so."""
        import torch.distributed as dist
        ids = np.zeros(lanes * _lib.COMM_ID_BYTES, dtype=np.uint8)
        if rank == 0:
            check(lib().impg_gpu_comm_unique_id(ids.ctypes.data, lanes))
        box = [ids.tobytes()]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        ids = np.frombuffer(box[0], dtype=np.uint8).copy()
        h = C.c_void_p(None)
        check(lib().impg_gpu_comm_create_rccl(ids.ctypes.data, lanes, rank, world, device, C.byref(h)))
        return cls(h, rank, world, lanes)

    @classmethod
    def host(cls, rank, world, device, groups):
        """The host's own transport (impg_gpu_comm_create_host): one torch.distributed group per lane, moving
        HOST memory (gloo).  What the multi-rank tests use on a box whose ranks share one GPU."""
        import torch
        import torch.distributed as dist
        lanes = len(groups)
        arr = (_lib.HostTransport * lanes)()
        keep = []

        def make(group):
            def allgather(ctx, mine, k, out):
                try:
                    t = torch.from_numpy(np.ctypeslib.as_array(mine, shape=(k,)).astype(np.int64))
                    parts = [torch.empty_like(t) for _ in range(world)]
                    dist.all_gather(parts, t, group=group)
                    dst = np.ctypeslib.as_array(out, shape=(world * k,))
                    for r in range(world):
                        dst[r * k:(r + 1) * k] = parts[r].numpy().astype(np.uint64)
                    return 0
                except Exception as e:  # noqa: BLE001 -- nothing may unwind into C
                    print("host transport allgather failed:", e, flush=True)
                    return 1

            def alltoallv(ctx, send, soff, sbytes, recv, roff, rbytes):
                try:
                    so = [int(soff[d]) for d in range(world)]
                    sb = [int(sbytes[d]) for d in range(world)]
                    ro = [int(roff[d]) for d in range(world)]
                    rb = [int(rbytes[d]) for d in range(world)]
                    st, rt = sum(sb), sum(rb)
                    src = (torch.from_numpy(np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(st,)).copy())
                           if st else torch.empty(0, dtype=torch.uint8))
                    dst = torch.empty(rt, dtype=torch.uint8)
                    assert so == list(np.cumsum([0] + sb[:-1])) and ro == list(np.cumsum([0] + rb[:-1]))
                    dist.all_to_all_single(dst, src, output_split_sizes=rb, input_split_sizes=sb, group=group)
                    if rt:
                        np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(rt,))[:] = dst.numpy()
                    return 0
                except Exception as e:  # noqa: BLE001
                    print("host transport alltoallv failed:", e, flush=True)
                    return 1
            return _lib.ALLGATHER_CB(allgather), _lib.ALLTOALLV_CB(alltoallv)

        for l, g in enumerate(groups):
            ag, aa = make(g)
            keep += [ag, aa]
            arr[l].ctx = None
            arr[l].allgather_u64 = ag
            arr[l].alltoallv = aa
        h = C.c_void_p(None)
        check(lib().impg_gpu_comm_create_host(arr, lanes, rank, world, device, C.byref(h)))
        return cls(h, rank, world, lanes, keep=(arr, keep))

    def check(self):
        """impg_gpu_comm_check on every lane (collective)."""
        for l in range(self.lanes):
            check(lib().impg_gpu_comm_check(self._h, l))

    def kind(self):
        k = C.c_char_p(None)
        check(lib().impg_gpu_comm_info(self._h, None, None, None, C.byref(k)))
        return k.value.decode()

    def close(self):
        if getattr(self, "_h", None):
            lib().impg_gpu_comm_destroy(self._h)
            self._h = None


def parse_subsequence(name):
    """impg_gpu_parse_subsequence: (base, start offset) of a "base:START-END" name, or None."""
    buf = C.create_string_buffer(len(name.encode()) + 2)
    off = C.c_int32(0)
    r = lib().impg_gpu_parse_subsequence(name.encode(), buf, len(buf), C.byref(off))
    if r < 0:
        check(r)
    return (buf.value.decode(), int(off.value)) if r == 1 else None


def subset_keep(list_text, names):
    """impg_gpu_subset_keep: (uint8 verdict per name, number of list entries) for a --subset-sequence-list text."""
    data = list_text.encode() if isinstance(list_text, str) else bytes(list_text)
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    keep = np.zeros(len(names), dtype=np.uint8)
    n_entries = C.c_size_t(0)
    check(lib().impg_gpu_subset_keep(data, len(data), arr, len(names), keep.ctypes.data, C.byref(n_entries)))
    return keep, int(n_entries.value)


def bed_merge(intervals, merge_distance, merge_strands=True):
    a = np.ascontiguousarray(intervals, dtype=INTERVAL_DTYPE).copy()
    n = lib().impg_gpu_bed_merge(a.ctypes.data, a.size, merge_distance, int(merge_strands))
    if n < 0:
        check(int(n))
    return a[:n].copy()


def parse_cigar(s):
    b = s.encode() if isinstance(s, str) else s
    out = np.zeros(len(b) + 1, dtype=np.uint32)
    n = lib().impg_gpu_parse_cigar(b, len(b), out.ctypes.data, out.size)
    if n < 0:
        raise ValueError("Invalid CIGAR operation")
    return out[:n].copy()


def parse_target_range(s):
    name = C.create_string_buffer(4096)
    a, b = C.c_int32(0), C.c_int32(0)
    rc = lib().impg_gpu_parse_target_range(s.encode(), name, 4096, C.byref(a), C.byref(b))
    if rc != 0:
        raise ValueError(lib().impg_gpu_last_error().decode())
    return name.value.decode(), a.value, b.value


# ---- synthetic workloads --------------------------------
def synth_paf(seed, n_records, n_seq=200, seq_len=5_000_000, target_span=10_000, n_blocks=100):
    n_ops = C.c_size_t(0)
    check(lib().impg_synth_paf(seed, n_records, n_seq, seq_len, target_span, n_blocks, None, None, 0, C.byref(n_ops)))
    rec = np.zeros(n_records, dtype=RECORD_DTYPE)
    ops = np.zeros(n_ops.value, dtype=np.uint32)
    check(lib().impg_synth_paf(seed, n_records, n_seq, seq_len, target_span, n_blocks, rec.ctypes.data, ops.ctypes.data,
                               ops.size, C.byref(n_ops)))
    return rec, ops, np.full(n_seq, seq_len, dtype=np.int64)


def synth_paf_text(path, seed, n_records, n_seq=200, seq_len=5_000_000, target_span=10_000, n_blocks=100):
    check(lib().impg_synth_paf_text(seed, n_records, n_seq, seq_len, target_span, n_blocks, path.encode()))
    return path


def synth_skewed_paf_text(path, seed, n_records, n_seq=200, seq_len=5_000_000):
    """impg_synth_skewed_paf_text: log-normal alignment lengths, 1 % of the sequences holding ~30 % of the entries; returns the ops written."""
    n = C.c_uint64(0)
    check(lib().impg_synth_skewed_paf_text(seed, n_records, n_seq, seq_len, path.encode(), C.byref(n)))
    return int(n.value)


def synth_seq_name(i):
    b = C.create_string_buffer(64)
    lib().impg_synth_seq_name(i, b, 64)
    return b.value.decode()


def synth_bed(seed, n, n_seq=200, seq_len=5_000_000, range_len=5_000):
    out = np.zeros(n, dtype=RANGE_DTYPE)
    check(lib().impg_synth_bed(seed, n, n_seq, seq_len, range_len, out.ctypes.data))
    return out
