"""impg_amd: MI355X-native batch interval-query + CIGAR-projection engine that
drops in behind impg's ImpgIndex::query / query_transitive_* (C ABI in
include/impg_gpu.h, HIP kernels in impg_amd/csrc)."""
from ._lib import (IMPG_E_HIP, IMPG_E_INVALID, IMPG_E_UNSUPPORTED, ORDER_COITREES, ORDER_SORTED, ImpgGpuError,
                   INTERVAL_DTYPE, RANGE_DTYPE, RECORD_DTYPE, build, lib)
from .index import Comm, DeviceRows, GpuImpg, PreparedMask, QueryResults, prepare_mask, shard_assign, make_params, parse_subsequence, subset_keep, synth_bed, synth_paf, synth_paf_text, synth_seq_name, synth_skewed_paf_text

__all__ = ["GpuImpg", "DeviceRows", "PreparedMask", "prepare_mask", "Comm", "shard_assign", "QueryResults", "make_params", "synth_paf", "synth_paf_text", "synth_skewed_paf_text", "synth_bed", "synth_seq_name",
           "build", "lib", "ImpgGpuError", "ORDER_COITREES", "ORDER_SORTED", "IMPG_E_HIP", "IMPG_E_INVALID",
           "IMPG_E_UNSUPPORTED", "INTERVAL_DTYPE", "RANGE_DTYPE", "RECORD_DTYPE"]
