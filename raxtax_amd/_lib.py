"""ctypes binding of include/raxtax_hip.h.  Fails loudly when libraxtax_hip.so is missing:
there is no Python or CPU fallback for the device path."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

PKG = Path(__file__).resolve().parent
import os as _os

LIB_PATH = Path(_os.environ.get("RTX_LIB_PATH") or PKG / "libraxtax_hip.so")   # RTX_LIB_PATH: an experimental build (tools/)

RTX_MAX_DEPTH = 32
RTX_NUM_KMERS = 65536
RTX_OK = 0
RTX_ERR_INVALID, RTX_ERR_HIP, RTX_ERR_NO_DEVICE, RTX_ERR_OOM = -1, -2, -3, -4
RTX_ERR_PARSE, RTX_ERR_DEPTH, RTX_ERR_STATE, RTX_ERR_TOO_LONG = -5, -6, -7, -8
RTX_SKIP_EXACT_MATCHES = 1
RTX_RAW_CONFIDENCE = 2
RTX_Q_OK, RTX_Q_NO_KMERS, RTX_Q_ALL_KMERS = 0, 1, 2
STAGES = ("kmer_extract", "hit_count", "prob_table", "taxon_prefix", "lineage_walk", "tile_bounds", "tile_prune", "exact_match", "order", "pair_union")

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f64p = C.POINTER(C.c_double)
f32p = C.POINTER(C.c_float)


class NodesView(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("node_begin", u32p), ("node_end", u32p), ("node_first_child", u32p),
                ("node_n_children", u32p), ("node_parent", u32p), ("node_type", u8p)]


class ResultView(C.Structure):
    _fields_ = [("n_queries", C.c_uint32), ("n_rows", C.c_uint64), ("t", u32p), ("status", u8p),
                ("global_signal", f64p), ("row_begin", u64p), ("row_count", u32p), ("row_lineage", u32p), ("row_node", u32p),
                ("row_depth", u32p), ("row_conf", f64p), ("row_local_signal", f64p),
                ("row_conf_stride", C.c_uint32), ("row_depth_u8", u8p), ("row_conf_hundredths", u8p)]   # ABI 5 (0 / NULL in a hand-made view)


class RtxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libraxtax_hip error {code}: {msg}")
        self.code = code


# every symbol include/raxtax_hip.h declares (tests/test_abi_symbols.py checks the export list)
_SIGNATURES = {
    "rtx_abi_version": (C.c_int, []),
    "rtx_last_error": (C.c_char_p, []),
    "rtx_device_count": (C.c_int, []),
    "rtx_tree_build": (C.c_int, [C.c_uint64, C.c_char_p, u64p, u8p, u64p, C.POINTER(C.c_void_p)]),
    "rtx_tree_build_ex": (C.c_int, [C.c_uint64, C.c_char_p, u64p, u8p, u64p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "rtx_tree_parse_reference_fasta": (C.c_int, [C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "rtx_tree_parse_reference_fasta_ex": (C.c_int, [C.c_char_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_void_p)]),
    "rtx_tree_build_kmer_map": (C.c_int, [C.c_void_p]),
    "rtx_tree_save_bin": (C.c_int, [C.c_void_p, C.c_char_p]),
    "rtx_tree_load_bin": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "rtx_tree_destroy": (None, [C.c_void_p]),
    "rtx_tree_num_tips": (C.c_uint64, [C.c_void_p]),
    "rtx_tree_lineage": (C.c_char_p, [C.c_void_p, C.c_uint64]),
    "rtx_tree_original_index": (C.c_uint64, [C.c_void_p, C.c_uint64]),
    "rtx_tree_kmer_csr": (C.c_int, [C.c_void_p, C.POINTER(u64p), C.POINTER(u32p)]),
    "rtx_tree_exact_matches": (C.c_uint64, [C.c_void_p, u8p, C.c_uint64, C.POINTER(u32p)]),
    "rtx_tree_exact_matches_batch": (C.c_uint64, [C.c_void_p, C.c_uint64, u8p, u64p, u64p, u32p, C.c_uint64]),
    "rtx_tree_nodes": (C.c_int, [C.c_void_p, C.POINTER(NodesView)]),
    "rtx_fasta_block_end": (C.c_uint64, [C.c_char_p, C.c_uint64]),
    "rtx_queries_parse_fasta_block": (C.c_int, [C.c_char_p, C.c_uint64, C.POINTER(C.c_char_p), C.c_uint64, C.c_uint32,
                                      C.POINTER(C.c_void_p)]),
    "rtx_queries_parse_fasta": (C.c_int, [C.c_char_p, C.c_uint64, C.POINTER(C.c_char_p), C.c_uint64,
                                          C.POINTER(C.c_void_p)]),
    "rtx_queries_destroy": (None, [C.c_void_p]),
    "rtx_queries_len": (C.c_uint64, [C.c_void_p]),
    "rtx_queries_label": (C.c_char_p, [C.c_void_p, C.c_uint64]),
    "rtx_queries_data": (C.c_int, [C.c_void_p, C.POINTER(u8p), C.POINTER(u64p)]),
    "rtx_index_create": (C.c_int, [C.c_int, C.c_uint64, u64p, u32p, C.c_uint32, u32p, u32p, u32p, u32p, u8p,
                                   C.POINTER(C.c_void_p)]),
    "rtx_index_create_shard": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, u64p, C.c_uint32, u64p, u32p, C.c_uint32,
                                        u32p, u32p, u32p, u32p, u8p, C.POINTER(C.c_void_p)]),
    "rtx_shard_begin": (C.c_int, [C.c_void_p, u32p, u32p]),
    "rtx_shard_count": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "rtx_shard_bounds": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "rtx_shard_prunes": (C.c_int, [C.c_void_p]),
    "rtx_shard_prob": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rtx_shard_walk": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "rtx_shard_info": (C.c_int, [C.c_void_p, u64p, u64p, u32p, u32p, u32p]),
    "rtx_device_buffer": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), u64p]),
    "rtx_shard_buffer": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_void_p), u64p]),
    "rtx_shard_rehist": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rtx_index_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "rtx_index_create_from_sequences": (C.c_int, [C.c_int, C.c_uint64, u8p, u64p, C.c_uint32, u32p, u32p, u32p, u32p, u8p,
                                                 C.POINTER(C.c_void_p)]),
    "rtx_index_create_from_tree": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "rtx_index_destroy": (None, [C.c_void_p]),
    "rtx_index_num_refs": (C.c_uint64, [C.c_void_p]),
    "rtx_index_device_bytes": (C.c_uint64, [C.c_void_p]),
    "rtx_index_workspace_bytes": (C.c_uint64, [C.c_void_p]),
    "rtx_index_workspace_parts": (C.c_int, [C.c_void_p, u64p]),
    "rtx_index_set_batch": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rtx_set_default_option": (C.c_int, [C.c_int, C.c_uint64]),
    "rtx_index_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64]),
    "rtx_index_self_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]),
    "rtx_index_prune_verdict": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "rtx_index_run_ahead_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rtx_classify_batch": (C.c_int, [C.c_void_p, C.c_uint64, u8p, u64p, u32p, u64p, C.c_uint32,
                                     C.POINTER(ResultView)]),
    "rtx_index_has_exact_lookup": (C.c_int, [C.c_void_p]),
    "rtx_batch_exact_matches": (C.c_int, [C.c_void_p, C.POINTER(u64p), C.POINTER(u32p)]),
    "rtx_batch_upload": (C.c_int, [C.c_void_p, C.c_uint64, u8p, u64p, u32p, u64p]),
    "rtx_batch_run": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rtx_batch_sync": (C.c_int, [C.c_void_p]),
    "rtx_batch_download": (C.c_int, [C.c_void_p, C.POINTER(ResultView)]),
    "rtx_batch_download_then_run": (C.c_int, [C.c_void_p, C.POINTER(ResultView), C.c_uint32]),
    "rtx_batch_stage_times": (C.c_int, [C.c_void_p, f32p, u32p]),
    "rtx_batch_work": (C.c_int, [C.c_void_p, u64p, u64p, u64p]),
    "rtx_batch_prob_work": (C.c_int, [C.c_void_p, u64p, u64p]),
    "rtx_batch_work_split": (C.c_int, [C.c_void_p, u64p, u64p]),
    "rtx_debug_kmers": (C.c_int, [C.c_void_p, C.c_uint64, u16p, u32p]),
    "rtx_debug_hit_counts": (C.c_int, [C.c_void_p, C.c_uint64, u16p]),
    "rtx_debug_prob_table": (C.c_int, [C.c_void_p, C.c_uint64, f64p, f64p]),
    "rtx_debug_pruned_prob_table": (C.c_int, [C.c_void_p, C.c_uint64, f64p, f64p, u32p]),
    "rtx_debug_probs": (C.c_int, [C.c_void_p, C.c_uint64, f64p]),
    "rtx_debug_order": (C.c_int, [C.c_void_p, u32p]),
    "rtx_debug_tile_bounds": (C.c_int, [C.c_void_p, C.c_uint64, u16p]),
    "rtx_debug_prune_stats": (C.c_int, [C.c_void_p, u64p]),
    "rtx_debug_run_counts": (C.c_int, [C.c_void_p, C.c_uint64, u16p, u8p, u32p, u32p, u32p]),
    "rtx_debug_run_mode": (C.c_int, [C.c_void_p, C.c_uint64, u32p]),
    "rtx_batch_last_sub_batch": (C.c_int, [C.c_void_p, u64p, u32p]),
    "rtx_batch_classes": (C.c_int, [C.c_void_p, u32p, u64p]),
    "rtx_debug_prune_detail": (C.c_int, [C.c_void_p, C.c_uint64, u32p]),
    "rtx_debug_evaluate": (C.c_int, [C.c_void_p, f64p, C.POINTER(ResultView)]),
    "rtx_result_pack": (C.c_int64, [C.POINTER(ResultView), u8p, C.c_uint64]),
    "rtx_records_format": (C.c_int64, [C.c_void_p, u8p, C.c_uint64, C.POINTER(C.c_char_p), u32p, C.c_uint32, C.c_char_p, C.c_uint64, u64p, C.c_uint32]),
    "rtx_format_query": (C.c_int64, [C.c_void_p, C.POINTER(ResultView), C.c_uint64, C.c_char_p, u8p, C.c_uint64,
                                     u32p, C.c_uint64, C.c_uint32, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64,
                                     C.POINTER(C.c_int64)]),
    "rtx_raxtax": (C.c_int, None),  # argtypes set in api.py (callback type)
    "rtx_raxtax_multi": (C.c_int, None),
    "rtx_sender_discard": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p]),
    "rtx_batch_prefetch": (C.c_int, [C.c_void_p, C.c_uint64, u8p, u64p, u32p, u64p]),
    "rtx_batch_activate": (C.c_int, [C.c_void_p]),
    "rtx_pack_bases": (C.c_int, [u8p, C.c_uint64, u8p]),
    "rtx_batch_sub_batch": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rtx_raxtax_last_timing": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "rtx_set_host_share": (C.c_int, [C.c_uint32]),
    "rtx_host_threads": (C.c_uint32, []),
}

_lib = None


def load() -> C.CDLL:
    """Loads libraxtax_hip.so (built by raxtax_amd._build.build_lib / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m raxtax_amd._build` (hipcc, gfx950). "
            "raxtax_amd has no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64; two HIP runtimes in one process do not both see the GPU.
    # Importing torch first makes libraxtax_hip.so resolve to the runtime that is already loaded, so the
    # library and torch.distributed (RCCL) share one (bench.py, raxtax_amd/sharded.py).  Optional.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing, not a requirement of the library
        pass
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = res
        if args is not None:
            fn.argtypes = args
    _lib = lib
    return lib


def check(code: int):
    if code < 0:
        raise RtxError(code, load().rtx_last_error().decode(errors="replace"))
    return code


def ptr(a: np.ndarray, typ):
    return a.ctypes.data_as(typ)
