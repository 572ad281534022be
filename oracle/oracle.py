"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the impg_amd package.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Interval(C.Structure):
    _fields_ = [("query_id", C.c_uint32), ("q_first", C.c_int32), ("q_last", C.c_int32),
                ("target_id", C.c_uint32), ("t_first", C.c_int32), ("t_last", C.c_int32)]


INTERVAL_DTYPE = np.dtype([("query_id", "<u4"), ("q_first", "<i4"), ("q_last", "<i4"),
                           ("target_id", "<u4"), ("t_first", "<i4"), ("t_last", "<i4")])


class Params(C.Structure):
    _fields_ = [("transitive", C.c_int32), ("dfs", C.c_int32), ("max_depth", C.c_uint32),
                ("min_transitive_len", C.c_int32), ("min_distance_between_ranges", C.c_int32),
                ("min_output_length", C.c_int32), ("min_identity", C.c_double),
                ("store_cigar", C.c_int32), ("multi_impg", C.c_int32), ("original_sequence_coordinates", C.c_int32),
                ("consider_strandness", C.c_int32)]


def make_params(transitive=False, dfs=False, max_depth=2, min_transitive_len=101,
                min_distance_between_ranges=10, min_output_length=None, min_identity=None,
                store_cigar=False, multi_impg=False, original_sequence_coordinates=False,
                consider_strandness=False):
    """Defaults are the reference CLI's (main.rs:4259-4285)."""
    return Params(int(transitive), int(dfs), max_depth, min_transitive_len,
                  min_distance_between_ranges,
                  -1 if min_output_length is None else min_output_length,
                  math.nan if min_identity is None else float(min_identity),
                  int(store_cigar), int(multi_impg), int(original_sequence_coordinates),
                  int(consider_strandness))


def parse_subsequence(name):
    """parse_subsequence_coordinates (main.rs:4642-4659): (base, offset) or None."""
    L = lib()
    L.oracle_parse_subsequence.restype = C.c_long
    L.oracle_parse_subsequence.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32)]
    buf = C.create_string_buffer(len(name.encode()) + 2)
    off = C.c_int32(0)
    r = L.oracle_parse_subsequence(name.encode(), buf, len(buf), C.byref(off))
    return (buf.value.decode(), int(off.value)) if r == 1 else None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("impg_oracle.cpp", "impg_oracle.h")]
    if force or not os.path.exists(so) or any(
            os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.oracle_last_error.restype = C.c_char_p
        L.oracle_last_projection_count.restype = C.c_uint64
        L.oracle_parse_cigar.restype = C.c_long
        L.oracle_parse_cigar.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.oracle_invert_cigar.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
        L.oracle_project.restype = C.c_int
        L.oracle_project.argtypes = [C.c_int32] * 6 + [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p,
                                                       C.c_void_p, C.POINTER(C.c_size_t)]
        L.oracle_gap_compressed_identity.restype = C.c_double
        L.oracle_gap_compressed_identity.argtypes = [C.c_void_p, C.c_size_t]
        L.oracle_sr_new.restype = C.c_void_p
        L.oracle_sr_new.argtypes = [C.c_int32, C.c_int32]
        L.oracle_sr_free.argtypes = [C.c_void_p]
        L.oracle_sr_insert.restype = C.c_long
        L.oracle_sr_insert.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t]
        L.oracle_sr_get.restype = C.c_long
        L.oracle_sr_get.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.oracle_index_from_paf.restype = C.c_void_p
        L.oracle_index_from_paf.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int]
        L.oracle_index_from_paf_text.restype = C.c_void_p
        L.oracle_index_from_paf_text.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int]
        L.oracle_index_free.argtypes = [C.c_void_p]
        L.oracle_num_seqs.restype = C.c_uint32
        L.oracle_num_seqs.argtypes = [C.c_void_p]
        L.oracle_seq_name.restype = C.c_char_p
        L.oracle_seq_name.argtypes = [C.c_void_p, C.c_uint32]
        L.oracle_seq_len.restype = C.c_int64
        L.oracle_seq_len.argtypes = [C.c_void_p, C.c_uint32]
        L.oracle_seq_id.restype = C.c_int64
        L.oracle_seq_id.argtypes = [C.c_void_p, C.c_char_p]
        L.oracle_num_records.restype = C.c_size_t
        L.oracle_num_records.argtypes = [C.c_void_p]
        L.oracle_num_targets.restype = C.c_size_t
        L.oracle_num_targets.argtypes = [C.c_void_p]
        L.oracle_target_entries.restype = C.c_size_t
        L.oracle_target_entries.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
        L.oracle_query.restype = C.c_long
        L.oracle_query.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_int32, C.POINTER(Params),
                                   C.c_void_p, C.c_size_t]
        L.oracle_query_masked.restype = C.c_long
        L.oracle_query_masked.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_int32, C.POINTER(Params), C.c_uint32,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.oracle_query_filtered.restype = C.c_long
        L.oracle_query_filtered.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_int32, C.POINTER(Params), C.c_int, C.c_uint32,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.oracle_query_cigar.restype = C.c_long
        L.oracle_query_cigar.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_int32, C.POINTER(Params), C.c_void_p,
                                         C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
        L.oracle_bed_merge.restype = C.c_long
        L.oracle_bed_merge.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_int]
        L.oracle_query_paf.restype = C.c_int
        L.oracle_query_paf.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.c_char_p,
                                       C.POINTER(Params), C.c_int32, C.c_int, C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.oracle_query_bed.restype = C.c_int
        L.oracle_query_bed.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.c_char_p,
                                       C.POINTER(Params), C.c_int32, C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.oracle_parse_target_range.restype = C.c_int
        L.oracle_parse_target_range.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t,
                                                C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.oracle_parse_bed_text.restype = C.c_long
        L.oracle_parse_bed_text.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_void_p,
                                            C.c_char_p, C.c_size_t, C.c_size_t]
        L.oracle_bench.restype = C.c_int
        L.oracle_bench.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                   C.POINTER(Params), C.c_int, C.c_int, C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        _LIB = L
    return _LIB


OPS = "=XIDM"


def op(length, ch):
    """CigarOp::new (impg.rs:81-93)"""
    return (OPS.index(ch) << 29) | length


def ops_from_pairs(pairs):
    return np.array([op(l, c) for l, c in pairs], dtype=np.uint32)


def ops_to_pairs(arr):
    return [(int(v) & ((1 << 29) - 1), OPS[int(v) >> 29]) for v in arr]


def parse_cigar(s):
    b = s.encode() if isinstance(s, str) else s
    out = np.zeros(len(b) + 1, dtype=np.uint32)
    n = lib().oracle_parse_cigar(b, len(b), out.ctypes.data, out.size)
    if n < 0:
        raise ValueError("Invalid CIGAR operation")
    return out[:n].copy()


def invert_cigar(ops, strand_reverse):
    a = np.ascontiguousarray(ops, dtype=np.uint32).copy()
    lib().oracle_invert_cigar(a.ctypes.data, a.size, int(strand_reverse))
    return a


def project(r, record, ops):
    """project_target_range_through_alignment; record=(ts,te,qs,qe,reverse).
    Returns None or (q_start,q_end,slice_ops,t_start,t_end)."""
    a = np.ascontiguousarray(ops, dtype=np.uint32)
    out4 = np.zeros(4, dtype=np.int32)
    sl = np.zeros(max(a.size, 1), dtype=np.uint32)
    n = C.c_size_t(0)
    ok = lib().oracle_project(r[0], r[1], record[0], record[1], record[2], record[3], int(record[4]),
                              a.ctypes.data, a.size, out4.ctypes.data, sl.ctypes.data, C.byref(n))
    if not ok:
        return None
    return int(out4[0]), int(out4[1]), sl[:n.value].copy(), int(out4[2]), int(out4[3])


def gap_compressed_identity(ops):
    a = np.ascontiguousarray(ops, dtype=np.uint32)
    return lib().oracle_gap_compressed_identity(a.ctypes.data, a.size)


class SortedRanges:
    def __init__(self, sequence_length, min_distance=0):
        self._h = lib().oracle_sr_new(sequence_length, min_distance)

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().oracle_sr_free(self._h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None

    def insert(self, a, b):
        out = np.zeros(2 * 4096, dtype=np.int32)
        n = lib().oracle_sr_insert(self._h, a, b, out.ctypes.data, 4096)
        return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n)]

    def ranges(self):
        out = np.zeros(2 * 65536, dtype=np.int32)
        n = lib().oracle_sr_get(self._h, out.ctypes.data, 65536)
        return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n)]


def subset_matches(list_text, names):
    """(uint8 verdict per name, entry_count) of the reference's SubsetFilter for a list file's text."""
    L = lib()
    L.oracle_subset_matches.restype = C.c_long
    L.oracle_subset_matches.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p]
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    out = np.zeros(len(names), dtype=np.uint8)
    n = L.oracle_subset_matches(list_text.encode(), arr, len(names), out.ctypes.data)
    return out, int(n)


def pack_mask(masked_regions):
    """{seq id: (sequence_length, [(start, end), ...])} -> the four flat arrays of the C interfaces
    (ids ascending, u64 CSR offsets, int32 start/end pairs)."""
    ids = sorted(masked_regions)
    mseq = np.array(ids, dtype=np.uint32)
    mlen = np.array([masked_regions[i][0] for i in ids], dtype=np.int32)
    moff = np.zeros(len(ids) + 1, dtype=np.uint64)
    flat = []
    for k, i in enumerate(ids):
        flat.extend(masked_regions[i][1])
        moff[k + 1] = len(flat)
    mrng = np.array(flat, dtype=np.int32).reshape(-1, 2) if flat else np.zeros((0, 2), dtype=np.int32)
    return mseq, mlen, moff, np.ascontiguousarray(mrng)


TP_RECORD_DTYPE = np.dtype([("query_id", "<u4"), ("target_id", "<u4"), ("query_start", "<i4"), ("query_end", "<i4"),
                            ("target_start", "<i4"), ("target_end", "<i4"), ("seg_off", "<u8"), ("n_segs", "<u4"),
                            ("strand", "<u4"), ("query_contig_start", "<i8")], align=True)
assert TP_RECORD_DTYPE.itemsize == 48


class OracleIndex:
    """Impg (and MultiImpg) built from PAF text or files."""

    def __init__(self, paf_text=None, paf_paths=None, bidirectional=True, preparse=False, tracepoints=None, impg_path=None):
        """tracepoints = dict(records=TP_RECORD_DTYPE[], tracepoints=int32[], query_deltas=int32[] | None (Standard),
        diffs=int32[] | None (FASTGA), fastga=bool, trace_spacing=int, max_complexity=int, seq_len=int64[]): an index over
        tracepoint alignments; every query on it runs in approximate mode (impg.rs:1317-1533)."""
        L = lib()
        if impg_path is not None:  # the reference's IMPGIDX2 file + the alignment files it was built from
            arr = (C.c_char_p * len(paf_paths))(*[p.encode() for p in paf_paths])
            L.oracle_index_from_impg.restype = C.c_void_p
            L.oracle_index_from_impg.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_int]
            self._h = L.oracle_index_from_impg(impg_path.encode(), arr, len(paf_paths), 0)
        elif tracepoints is not None:
            t = tracepoints
            rec = np.ascontiguousarray(t["records"], dtype=TP_RECORD_DTYPE)
            tp = np.ascontiguousarray(t["tracepoints"], dtype=np.int32)
            qd = None if t.get("query_deltas") is None else np.ascontiguousarray(t["query_deltas"], dtype=np.int32)
            df = None if t.get("diffs") is None else np.ascontiguousarray(t["diffs"], dtype=np.int32)
            sl = np.ascontiguousarray(t["seq_len"], dtype=np.int64)
            L.oracle_index_from_tracepoints.restype = C.c_void_p
            L.oracle_index_from_tracepoints.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int32,
                                                        C.c_int32, C.c_void_p, C.c_uint32, C.c_int]
            self._h = L.oracle_index_from_tracepoints(rec.ctypes.data, rec.size, tp.ctypes.data,
                                                      None if qd is None else qd.ctypes.data, None if df is None else df.ctypes.data,
                                                      int(bool(t.get("fastga"))), int(t.get("trace_spacing", 0)),
                                                      int(t.get("max_complexity", 0)), sl.ctypes.data, sl.size, int(bidirectional))
        elif paf_text is not None:
            b = paf_text.encode() if isinstance(paf_text, str) else bytes(paf_text)
            self._h = L.oracle_index_from_paf_text(b, len(b), int(bidirectional), int(preparse))
        else:
            arr = (C.c_char_p * len(paf_paths))(*[p.encode() for p in paf_paths])
            self._h = L.oracle_index_from_paf(arr, len(paf_paths), int(bidirectional), int(preparse))
        if not self._h:
            raise RuntimeError(L.oracle_last_error().decode())

    def write_impg(self, path, shuffle_seed=0):
        """oracle_index_write_impg: this index as the reference's IMPGIDX2 file."""
        f = lib().oracle_index_write_impg
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
        if f(self._h, path.encode(), shuffle_seed) != 0:
            raise RuntimeError(lib().oracle_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().oracle_index_free(self._h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None

    @property
    def handle(self):
        return self._h

    def num_seqs(self):
        return lib().oracle_num_seqs(self._h)

    def seq_name(self, i):
        return lib().oracle_seq_name(self._h, i).decode()

    def seq_len(self, i):
        return lib().oracle_seq_len(self._h, i)

    def seq_id(self, name):
        r = lib().oracle_seq_id(self._h, name.encode())
        return None if r < 0 else int(r)

    def num_records(self):
        return lib().oracle_num_records(self._h)

    def num_targets(self):
        return lib().oracle_num_targets(self._h)

    def target_entries(self, target_id):
        n = lib().oracle_target_entries(self._h, target_id, None, 0)
        out = np.zeros((max(n, 1), 4), dtype=np.int32)
        lib().oracle_target_entries(self._h, target_id, out.ctypes.data, n)
        return out[:n]

    def query(self, target_id, start, end, params=None, masked_regions=None, subset_keep=None, **kw):
        """Results (numpy structured array) in reference emission order.
        masked_regions: {sequence id: (sequence_length, [(start, end), ...])} -- the
        Option<&FxHashMap<u32, SortedRanges>> of query_transitive_{bfs,dfs}."""
        p = params or make_params(**kw)
        cap = 1 << 12
        if masked_regions is not None:
            mseq, mlen, moff, mrng = pack_mask(masked_regions)
        if subset_keep is not None:  # subset filter: keep[id] = SubsetFilter::matches(name of id), decided by the caller
            keep = np.ascontiguousarray(subset_keep, dtype=np.uint8)
            assert keep.size == self.num_seqs()
        while True:
            out = np.zeros(cap, dtype=INTERVAL_DTYPE)
            if subset_keep is not None:
                if masked_regions is not None:
                    n = lib().oracle_query_filtered(self._h, target_id, start, end, C.byref(p), 1, len(mseq), mseq.ctypes.data,
                                                    mlen.ctypes.data, moff.ctypes.data, mrng.ctypes.data, keep.ctypes.data,
                                                    out.ctypes.data, cap)
                else:
                    n = lib().oracle_query_filtered(self._h, target_id, start, end, C.byref(p), 0, 0, None, None, None, None,
                                                    keep.ctypes.data, out.ctypes.data, cap)
            elif masked_regions is not None:
                n = lib().oracle_query_masked(self._h, target_id, start, end, C.byref(p), len(mseq), mseq.ctypes.data,
                                              mlen.ctypes.data, moff.ctypes.data, mrng.ctypes.data, out.ctypes.data, cap)
            else:
                n = lib().oracle_query(self._h, target_id, start, end, C.byref(p), out.ctypes.data, cap)
            if n < 0:
                raise RuntimeError(lib().oracle_last_error().decode())
            if n <= cap:
                return out[:n].copy()
            cap = n

    def query_cigar(self, target_id, start, end, params=None, **kw):
        """(results, [ops array per result]) with store_cigar."""
        p = params or make_params(**kw)
        cap, ops_cap = 1 << 12, 1 << 18
        while True:
            out = np.zeros(cap, dtype=INTERVAL_DTYPE)
            off = np.zeros(cap + 1, dtype=np.uint64)
            ops = np.zeros(ops_cap, dtype=np.uint32)
            nops = C.c_uint64(0)
            n = lib().oracle_query_cigar(self._h, target_id, start, end, C.byref(p), out.ctypes.data, cap,
                                         off.ctypes.data, ops.ctypes.data, ops_cap, C.byref(nops))
            if n < 0:
                raise RuntimeError(lib().oracle_last_error().decode())
            if n <= cap and nops.value <= ops_cap:
                return out[:n].copy(), [ops[int(off[i]):int(off[i + 1])].copy() for i in range(n)]
            cap, ops_cap = max(cap, n), max(ops_cap, int(nops.value))

    def last_projection_count(self):
        return lib().oracle_last_projection_count()

    def query_bed(self, target_name, start, end, range_name=None, merge_distance=0, params=None, **kw):
        p = params or make_params(**kw)
        if range_name is None:
            range_name = "%s:%d-%d" % (target_name, start, end)
        buf = C.c_void_p(None)
        ln = C.c_size_t(0)
        cap = C.c_size_t(0)
        rc = lib().oracle_query_bed(self._h, target_name.encode(), start, end, range_name.encode(),
                                    C.byref(p), merge_distance, C.byref(buf), C.byref(ln), C.byref(cap))
        try:
            if rc != 0:
                raise RuntimeError(lib().oracle_last_error().decode())
            return C.string_at(buf, ln.value).decode() if ln.value else ""
        finally:
            if buf.value:
                C.CDLL(None).free(buf)

    def query_paf(self, target_name, start, end, range_name=None, merge_distance=0, fmt="paf", params=None, **kw):
        """`impg query -o paf|bedpe` text for one target range."""
        p = params or make_params(**kw)
        if range_name is None:
            range_name = "%s:%d-%d" % (target_name, start, end)
        buf = C.c_void_p(None)
        ln = C.c_size_t(0)
        cap = C.c_size_t(0)
        rc = lib().oracle_query_paf(self._h, target_name.encode(), start, end, range_name.encode(), C.byref(p),
                                    merge_distance, {"paf": 0, "bedpe": 1}[fmt], C.byref(buf), C.byref(ln), C.byref(cap))
        try:
            if rc != 0:
                raise RuntimeError(lib().oracle_last_error().decode())
            return C.string_at(buf, ln.value).decode() if ln.value else ""
        finally:
            if buf.value:
                C.CDLL(None).free(buf)

    def bench(self, target_ids, starts, ends, params, threads=1, mode=0):
        t = np.ascontiguousarray(target_ids, dtype=np.uint32)
        s = np.ascontiguousarray(starts, dtype=np.int32)
        e = np.ascontiguousarray(ends, dtype=np.int32)
        npj = C.c_uint64(0)
        nr = C.c_uint64(0)
        sec = C.c_double(0)
        rc = lib().oracle_bench(self._h, t.ctypes.data, s.ctypes.data, e.ctypes.data, t.size,
                                C.byref(params), threads, mode, C.byref(npj), C.byref(nr), C.byref(sec))
        if rc != 0:
            raise RuntimeError(lib().oracle_last_error().decode())
        return npj.value, nr.value, sec.value


def set_sorted_visits(on):
    """oracle_set_sorted_visits: process-wide; the checker of the engine's IMPG_ORDER_SORTED policy."""
    lib().oracle_set_sorted_visits(int(bool(on)))


def merge_query(intervals, merge_distance, merge_strands=True):
    a = np.ascontiguousarray(intervals, dtype=INTERVAL_DTYPE).copy()
    f = lib().oracle_merge_query
    f.restype = C.c_long
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_int]
    n = f(a.ctypes.data, a.size, merge_distance, int(merge_strands))
    return a[:n].copy()


def bed_merge(intervals, merge_distance, merge_strands=True):
    a = np.ascontiguousarray(intervals, dtype=INTERVAL_DTYPE).copy()
    n = lib().oracle_bed_merge(a.ctypes.data, a.size, merge_distance, int(merge_strands))
    return a[:n].copy()


def parse_target_range(s):
    name = C.create_string_buffer(4096)
    a = C.c_int32(0)
    b = C.c_int32(0)
    rc = lib().oracle_parse_target_range(s.encode(), name, 4096, C.byref(a), C.byref(b))
    if rc != 0:
        raise ValueError("bad target range (%d)" % rc)
    return name.value.decode(), a.value, b.value


def parse_bed_text(text):
    b = text.encode() if isinstance(text, str) else text
    cap = b.count(b"\n") + 2
    names = C.create_string_buffer(len(b) + 16)
    rn = C.create_string_buffer(len(b) * 2 + 64 * cap)
    se = np.zeros(2 * cap, dtype=np.int32)
    n = lib().oracle_parse_bed_text(b, len(b), names, len(names), se.ctypes.data, rn, len(rn), cap)
    if n < 0:
        raise ValueError("Invalid BED file format (%d)" % n)
    ns = names.value.decode().split("\n")[:n]
    rs = rn.value.decode().split("\n")[:n]
    return [(ns[i], int(se[2 * i]), int(se[2 * i + 1]), rs[i]) for i in range(n)]
