"""ctypes loader for libimpg_gpu.so (the C ABI of include/impg_gpu.h).

The library is built in-tree by `make -C impg_amd/csrc` (hipcc, gfx950); there
is no CPU fallback: a missing library is an ImportError-like RuntimeError, and
every query on a machine without a GPU fails with IMPG_E_HIP.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IMPG_GPU_LIB") or os.path.join(_HERE, "libimpg_gpu.so")  # override: kernel experiments
CSRC = os.path.join(_HERE, "csrc")

IMPG_OK = 0
IMPG_E_INVALID, IMPG_E_HIP, IMPG_E_OOM, IMPG_E_IO, IMPG_E_UNSUPPORTED, IMPG_E_CANCELLED = -1, -2, -3, -4, -5, -6
STREAM_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
ORDER_COITREES, ORDER_SORTED = 0, 1
HIT_NONE = 0xFFFFFFFF

RECORD_DTYPE = np.dtype([("query_id", "<u4"), ("target_id", "<u4"), ("query_start", "<i4"), ("query_end", "<i4"),
                         ("target_start", "<i4"), ("target_end", "<i4"), ("cigar_off", "<u8"), ("cigar_len", "<u4"),
                         ("strand", "<u4")], align=True)
RANGE_DTYPE = np.dtype([("target_id", "<u4"), ("start", "<i4"), ("end", "<i4")])
INTERVAL_DTYPE = np.dtype([("query_id", "<u4"), ("q_first", "<i4"), ("q_last", "<i4"),
                           ("target_id", "<u4"), ("t_first", "<i4"), ("t_last", "<i4")])
assert RECORD_DTYPE.itemsize == 40 and RANGE_DTYPE.itemsize == 12
TP_RECORD_DTYPE = np.dtype([("query_id", "<u4"), ("target_id", "<u4"), ("query_start", "<i4"), ("query_end", "<i4"),
                            ("target_start", "<i4"), ("target_end", "<i4"), ("seg_off", "<u8"), ("n_segs", "<u4"),
                            ("strand", "<u4"), ("query_contig_start", "<i8")], align=True)  # impg_gpu_tp_record_t
assert TP_RECORD_DTYPE.itemsize == 48
COMM_ID_BYTES = 128


class TpMode(C.Structure):  # impg_gpu_tp_mode_t
    _fields_ = [("fastga", C.c_int32), ("trace_spacing", C.c_int32), ("max_complexity", C.c_int32)]


ALLGATHER_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_uint64))
ALLTOALLV_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p,
                           C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))


class HostTransport(C.Structure):  # impg_gpu_host_transport_t
    _fields_ = [("ctx", C.c_void_p), ("allgather_u64", ALLGATHER_CB), ("alltoallv", ALLTOALLV_CB)]


class Params(C.Structure):
    _fields_ = [("transitive", C.c_int32), ("dfs", C.c_int32), ("max_depth", C.c_uint32),
                ("min_transitive_len", C.c_int32), ("min_distance_between_ranges", C.c_int32),
                ("min_output_length", C.c_int32), ("min_identity", C.c_double),
                ("store_cigar", C.c_int32), ("multi_impg", C.c_int32), ("original_sequence_coordinates", C.c_int32),
                ("consider_strandness", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("projected", C.c_uint64), ("pairs", C.c_uint64), ("frontier_ranges", C.c_uint64),
                ("levels", C.c_uint32), ("ms_total", C.c_float), ("ms_lookup", C.c_float),
                ("ms_project", C.c_float), ("ms_update", C.c_float), ("project_launches", C.c_uint64),
                ("ms_exchange", C.c_float)]


class DevicePart(C.Structure):  # impg_gpu_device_part_t
    _fields_ = [("first_range", C.c_size_t), ("n_ranges", C.c_size_t), ("level", C.c_uint32), ("n_frontier", C.c_uint32),
                ("slot_stride", C.c_uint32), ("n_slots", C.c_uint64), ("query_id", C.c_void_p), ("coords", C.c_void_p), ("source", C.c_void_p),
                ("frontier", C.c_void_p), ("rows", C.c_void_p), ("offsets", C.c_void_p)]


ROWS_ATTRIBUTED, ROWS_ORDERED, ROWS_ORDERED_SLOTS = 0, 1, 2
FRONTIER_DTYPE = np.dtype([("target_id", "<u4"), ("start", "<i4"), ("end", "<i4"), ("range_idx", "<u4")])


class ImpgGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("impg_gpu error %d: %s" % (code, msg))
        self.code = code


def build(force=False):
    """Compile libimpg_gpu.so for gfx950 (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
    return LIB_PATH


class Mask(C.Structure):  # impg_gpu_mask_t
    _fields_ = [("n_seqs", C.c_uint32), ("seq_id", C.c_void_p), ("sequence_length", C.c_void_p),
                ("range_off", C.c_void_p), ("ranges", C.c_void_p)]


# every symbol include/impg_gpu.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("impg_gpu_last_error", C.c_char_p, []),
    ("impg_gpu_device_count", C.c_int, []),
    ("impg_gpu_index_create", C.c_int, [_P, C.c_size_t, _P, C.c_size_t, _P, C.c_uint32, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    ("impg_gpu_index_create_files", C.c_int, [_P, C.c_size_t, _P, C.c_size_t, _P, C.c_uint32, _P, C.c_uint32, C.c_int, C.c_int,
                                              C.c_int, C.POINTER(_P)]),
    ("impg_gpu_index_create_tracepoints", C.c_int, [_P, C.c_size_t, _P, _P, _P, C.c_size_t, _P, _P, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                                    C.POINTER(_P)]),
    ("impg_gpu_index_create_tracepoints_multi", C.c_int, [_P, C.c_size_t, _P, _P, _P, C.c_size_t, _P, _P, C.c_uint32, C.c_int, C.c_int, _P, C.c_int, C.c_int,
                                                    C.POINTER(_P)]),
    ("impg_gpu_index_create_from_paf", C.c_int, [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    ("impg_gpu_index_save", C.c_int, [_P, C.c_char_p]),
    ("impg_gpu_index_load", C.c_int, [C.c_char_p, C.c_int, C.POINTER(_P)]),
    ("impg_gpu_query_batch_stream", C.c_int, [_P, _P, C.c_size_t, _P, _P, _P, C.c_size_t, C.c_size_t, _P, _P, C.POINTER(C.c_uint64)]),
    ("impg_gpu_index_load_rank", C.c_int, [C.c_char_p, C.c_int, _P, C.POINTER(_P)]),
    ("impg_gpu_index_load_multi", C.c_int, [C.c_char_p, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    ("impg_gpu_index_load_impg", C.c_int, [C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    ("impg_gpu_index_destroy", None, [_P]),
    ("impg_gpu_num_seqs", C.c_uint32, [_P]),
    ("impg_gpu_seq_name", C.c_char_p, [_P, C.c_uint32]),
    ("impg_gpu_seq_len", C.c_int64, [_P, C.c_uint32]),
    ("impg_gpu_seq_id", C.c_int64, [_P, C.c_char_p]),
    ("impg_gpu_num_targets", C.c_size_t, [_P]),
    ("impg_gpu_target_ids", C.c_size_t, [_P, _P, C.c_size_t]),
    ("impg_gpu_num_entries", C.c_size_t, [_P]),
    ("impg_gpu_num_records", C.c_size_t, [_P]),
    ("impg_gpu_device_bytes", C.c_size_t, [_P]),
    ("impg_gpu_index_approximate", C.c_int, [_P]),
    ("impg_gpu_set_option", C.c_int, [_P, C.c_char_p, C.c_int64]),
    ("impg_gpu_get_counter", C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    ("impg_gpu_host_pool_trim", C.c_uint64, [C.c_uint64]),
    ("impg_gpu_visit_rank", C.c_int, [C.c_uint32, C.c_int, _P]),
    ("impg_gpu_query_batch", C.c_int, [_P, _P, C.c_size_t, C.POINTER(Params), C.POINTER(_P)]),
    ("impg_gpu_query_batch_masked", C.c_int, [_P, _P, C.c_size_t, C.POINTER(Params), _P, C.POINTER(_P)]),
    ("impg_gpu_query_batch_filtered", C.c_int, [_P, _P, C.c_size_t, C.POINTER(Params), _P, _P, C.POINTER(_P)]),
    ("impg_gpu_parse_subsequence", C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32)]),
    ("impg_gpu_subset_keep", C.c_int, [C.c_char_p, C.c_size_t, _P, C.c_size_t, _P, C.POINTER(C.c_size_t)]),
    ("impg_gpu_query", C.c_int, [_P, C.c_uint32, C.c_int32, C.c_int32, C.POINTER(Params), C.POINTER(_P)]),
    ("impg_gpu_results_num_ranges", C.c_size_t, [_P]),
    ("impg_gpu_results_total", C.c_size_t, [_P]),
    ("impg_gpu_results_offsets", _P, [_P]),
    ("impg_gpu_results_intervals", _P, [_P]),
    ("impg_gpu_results_projected", C.c_uint64, [_P]),
    ("impg_gpu_results_cigar_offsets", _P, [_P]),
    ("impg_gpu_results_cigar_ops", _P, [_P]),
    ("impg_gpu_results_timing", None, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("impg_gpu_results_free", None, [_P]),
    ("impg_gpu_query_batch_stats", C.c_int, [_P, _P, C.c_size_t, C.POINTER(Params), _P, _P, C.POINTER(Stats)]),
    ("impg_gpu_query_batch_stats_dev", C.c_int, [_P, _P, C.c_size_t, C.POINTER(Params), _P, _P, C.POINTER(Stats)]),
    ("impg_gpu_query_batch_device", C.c_int, [_P, _P, C.c_size_t, C.c_int, C.POINTER(Params), C.c_int, C.POINTER(_P)]),
    ("impg_gpu_device_rows_num_parts", C.c_size_t, [_P]),
    ("impg_gpu_device_rows_part", C.c_int, [_P, C.c_size_t, C.POINTER(DevicePart)]),
    ("impg_gpu_device_rows_stats", None, [_P, C.POINTER(Stats)]),
    ("impg_gpu_device_rows_place_ms", C.c_float, [_P]),
    ("impg_gpu_device_rows_check", C.c_int, [_P, _P, _P]),
    ("impg_gpu_device_rows_free", None, [_P]),
    ("impg_gpu_bed_merge", C.c_long, [_P, C.c_size_t, C.c_int32, C.c_int]),
    ("impg_gpu_results_bed", C.c_int, [_P, _P, _P, C.POINTER(Params), C.c_int32, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    ("impg_gpu_query_batch_bed", C.c_int, [_P, _P, C.c_size_t, C.POINTER(Params), _P, C.c_int32, _P, C.POINTER(_P), C.POINTER(C.c_size_t),
                                           _P]),
    ("impg_gpu_query_batch_bed_fd", C.c_int, [_P, _P, C.c_size_t, C.POINTER(Params), _P, C.c_int32, _P, C.c_int, C.POINTER(C.c_uint64), _P]),
    ("impg_gpu_results_paf", C.c_int, [_P, _P, _P, C.POINTER(Params), C.c_int32, C.c_int, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    ("impg_gpu_parse_cigar", C.c_long, [C.c_char_p, C.c_size_t, _P, C.c_size_t]),
    ("impg_gpu_parse_target_range", C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("impg_gpu_comm_unique_id", C.c_int, [_P, C.c_int]),
    ("impg_gpu_comm_create_rccl", C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    ("impg_gpu_comm_create_host", C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    ("impg_gpu_comm_destroy", None, [_P]),
    ("impg_gpu_comm_info", C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p)]),
    ("impg_gpu_comm_check", C.c_int, [_P, C.c_int]),
    ("impg_gpu_index_create_rank", C.c_int, [_P, C.c_size_t, _P, C.c_size_t, _P, C.c_uint32, _P, C.c_uint32, C.c_int, C.c_int,
                                             C.c_int, _P, C.POINTER(_P)]),
    ("impg_gpu_index_create_from_paf_rank", C.c_int, [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, _P, C.POINTER(_P)]),
    ("impg_gpu_index_create_multi", C.c_int, [_P, C.c_size_t, _P, C.c_size_t, _P, C.c_uint32, _P, C.c_uint32, C.c_int, C.c_int,
                                              _P, C.c_int, C.c_int, C.POINTER(_P)]),
    ("impg_gpu_index_create_from_paf_multi", C.c_int, [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int,
                                                       C.POINTER(_P)]),
    ("impg_gpu_shard_assign", C.c_int, [_P, C.c_uint32, C.c_uint32, _P]),
    ("impg_gpu_index_shard_info", C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), _P, C.c_size_t]),
    ("impg_gpu_index_hop_profile", C.c_int, [_P, _P, C.c_size_t, C.c_int, C.POINTER(C.c_size_t)]),
    ("impg_synth_paf", C.c_int, [C.c_uint64, C.c_size_t, C.c_uint32, C.c_int32, C.c_int32, C.c_uint32, _P, _P, C.c_size_t,
                                 C.POINTER(C.c_size_t)]),
    ("impg_synth_paf_text", C.c_int, [C.c_uint64, C.c_size_t, C.c_uint32, C.c_int32, C.c_int32, C.c_uint32, C.c_char_p]),
    ("impg_synth_skewed_paf_text", C.c_int, [C.c_uint64, C.c_size_t, C.c_uint32, C.c_int32, C.c_char_p, C.POINTER(C.c_uint64)]),
    ("impg_gpu_selftest_order_sort", C.c_int, [C.c_int, C.c_uint32, C.c_uint, C.c_uint64]),
    ("impg_synth_seq_name", C.c_int, [C.c_uint32, C.c_char_p, C.c_size_t]),
    ("impg_synth_bed", C.c_int, [C.c_uint64, C.c_size_t, C.c_uint32, C.c_int32, C.c_int32, _P]),
]

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libimpg_gpu.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "or `make -C impg_amd/csrc` (there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB


def check(rc):
    if rc != IMPG_OK:
        raise ImpgGpuError(rc, lib().impg_gpu_last_error().decode(errors="replace"))


def free(ptr):
    C.CDLL(None).free(ptr)
