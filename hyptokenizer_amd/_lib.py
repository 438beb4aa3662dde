"""ctypes binding of libhypmerge.so (C ABI declared in include/hypmerge.h).

There is no CPU fallback: if the shared library is missing or fails to load, importing the
engine raises ``HypMergeUnavailable`` with build instructions.
"""
from __future__ import annotations

import ctypes as C
import os

# torch must come first: it ships its own HIP runtime (libamdhip64) and libhypmerge.so has to bind
# to that one.  Loading our library first would pull in the system runtime and leave the process
# with two HIP runtimes (the second one then reports "no device").
import torch  # noqa: F401  (side effect: loads torch's libamdhip64)

_HERE = os.path.dirname(os.path.abspath(__file__))
# HYPMERGE_LIB: load another build of the same library (tools/ run kernel variants side by side); there is still no
# fallback -- a path that does not load raises like the default one
LIB_PATH = os.environ.get("HYPMERGE_LIB") or os.path.join(_HERE, "libhypmerge.so")

HM_OK = 0
HM_E_ARG, HM_E_CAPACITY, HM_E_STATE, HM_E_NOMEM, HM_E_COMM, HM_E_NA = -1, -2, -3, -4, -5, -6
COMM_ID_BYTES = 128
SIGN_REFERENCE, SIGN_LORENTZ = 0, 1
PREFILTER_AUTO, PREFILTER_F32, PREFILTER_BF16 = 0, 1, 2
LOOP_MAX_STEPS = 256

#: every symbol include/hypmerge.h declares (tests check that the library exports all of them)
EXPORTED_SYMBOLS = (
    "hm_abi_version", "hm_last_error", "hm_engine_create", "hm_engine_destroy", "hm_set_table",
    "hm_update_rows", "hm_rows", "hm_pairwise_argmin", "hm_pairwise_argmin_dev", "hm_pairwise_topk", "hm_pairwise_candidates",
    "hm_row_vs_all", "hm_row_argmin", "hm_pair_distance", "hm_midpoint_batch", "hm_merge_append", "hm_batch_distance",
    "hm_rows_minkowski", "hm_rows_distance", "hm_rows_log_map", "hm_rows_exp_map", "hm_rows_project",
    "hm_last_scan_stats", "hm_scan_totals", "hm_set_prefilter", "hm_pairwise_topk_nocount", "hm_pairwise_count",
    "hm_merge_append_batch", "hm_truncate", "hm_set_token_lengths", "hm_std_merge_steps", "hm_incr_merge_steps",
    "hm_coherence_batch", "hm_project_table", "hm_debug_force_cut", "hm_randperm_prefix", "hm_merge_append_batch_host",
    "hm_tokenize_table_capacity", "hm_tokenize_build_table", "hm_tokenize_batch", "hm_debug_time_loops", "hm_last_loop_timing", "hm_shard_loop_begin", "hm_shard_merge_step", "hm_shard_loop_end",
    "hm_topk_refresh_begin", "hm_topk_refresh_end", "hm_debug_set_knob", "hm_debug_set_default_knob",
    "hm_comm_unique_id", "hm_comm_init", "hm_comm_destroy", "hm_comm_info", "hm_shard_merge_steps", "hm_global_argmin", "hm_global_topk",
)


class HypMergeUnavailable(RuntimeError):
    """libhypmerge.so could not be loaded (it must be built with hipcc for gfx950)."""


class HypMergeError(RuntimeError):
    """A C-ABI call returned a non-zero status."""

    def __init__(self, status: int, message: str):
        super().__init__(f"libhypmerge status {status}: {message}")
        self.status = status


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HypMergeUnavailable(
            f"{LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C hyptokenizer_amd/csrc` (needs hipcc, --offload-arch=gfx950). "
            "The merge engine has no CPU fallback.")
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as exc:  # missing ROCm runtime, wrong arch, ...
        raise HypMergeUnavailable(f"cannot load {LIB_PATH}: {exc}") from exc

    vp, f32, i32, i64 = C.c_void_p, C.c_float, C.c_int32, C.c_int64
    pf32, pi32, pi64 = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64)

    L.hm_abi_version.restype = C.c_int
    L.hm_last_error.restype = C.c_char_p
    L.hm_last_error.argtypes = [vp]
    L.hm_engine_create.argtypes = [C.POINTER(vp), C.c_int, i64, C.c_int, C.c_int, C.c_int]
    L.hm_set_prefilter.argtypes = [vp, C.c_int]
    L.hm_pairwise_topk_nocount.argtypes = [vp, f32, f32, i64, i64, i64, vp, vp, vp, pi64, pi64, vp]
    L.hm_pairwise_count.argtypes = [vp, f32, f32, i64, pi64, vp]
    L.hm_merge_append_batch.argtypes = [vp, vp, vp, vp, i64, f32, vp, i64, i64, C.c_int, vp]
    L.hm_merge_append_batch_host.argtypes = [vp, vp, vp, vp, i64, f32, vp, i64, i64, C.c_int, vp]
    L.hm_truncate.argtypes = [vp, i64, vp]
    L.hm_set_token_lengths.argtypes = [vp, vp, i64, vp]
    L.hm_std_merge_steps.argtypes = [vp, f32, f32, vp, i64, i64, vp, pi64, vp]
    L.hm_incr_merge_steps.argtypes = [vp, f32, f32, vp, i64, i64, vp, vp, pi64, vp]
    L.hm_coherence_batch.argtypes = [vp, vp, vp, vp, vp, i64, C.c_int, f32, vp, vp]
    L.hm_project_table.argtypes = [vp, vp, i64, i64, f32, vp]
    L.hm_debug_force_cut.argtypes = [vp, C.c_uint32, i64, f32]
    L.hm_randperm_prefix.argtypes = [vp, pi32, C.POINTER(C.c_uint32), i64, i32, i64, vp]
    L.hm_debug_time_loops.argtypes = [vp, C.c_int]
    L.hm_shard_loop_begin.argtypes = [vp, vp]
    L.hm_topk_refresh_begin.argtypes = [vp, f32, f32, i64, vp]
    L.hm_topk_refresh_end.argtypes = [vp, vp, vp, vp, pi64]
    L.hm_shard_merge_step.argtypes = [vp, vp, C.c_int, f32, vp, i64, i64, vp]
    L.hm_shard_loop_end.argtypes = [vp, i64, vp, pi64, vp]
    L.hm_last_loop_timing.argtypes = [vp, pf32, pf32, pi64]
    L.hm_tokenize_table_capacity.restype = i64
    L.hm_tokenize_table_capacity.argtypes = [i64]
    L.hm_tokenize_build_table.argtypes = [vp, vp, vp, i64, vp, i64]
    L.hm_tokenize_batch.argtypes = [vp, vp, vp, i64, vp, i64, vp, vp, vp, vp]
    L.hm_comm_unique_id.argtypes = [vp]
    L.hm_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    L.hm_comm_destroy.argtypes = [vp]
    L.hm_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.hm_shard_merge_steps.argtypes = [vp, f32, f32, vp, i64, i64, vp, pi64, vp]
    L.hm_global_argmin.argtypes = [vp, f32, f32, pf32, pi32, pi32, pi32, vp]
    L.hm_global_topk.argtypes = [vp, f32, f32, i64, vp, vp, vp, pi64, pi64, vp]
    L.hm_debug_set_knob.argtypes = [vp, C.c_char_p, C.c_double]
    L.hm_debug_set_default_knob.argtypes = [C.c_char_p, C.c_double, C.c_int]
    L.hm_engine_destroy.argtypes = [vp]
    L.hm_set_table.argtypes = [vp, vp, i64, i64, vp]
    L.hm_update_rows.argtypes = [vp, vp, i64, i64, i64, vp]
    L.hm_rows.restype = i64
    L.hm_rows.argtypes = [vp]
    L.hm_pairwise_argmin.argtypes = [vp, f32, f32, i64, i64, pf32, pi32, pi32, pi32, vp]
    L.hm_pairwise_argmin_dev.argtypes = [vp, f32, f32, i64, i64, vp, vp]
    L.hm_pairwise_topk.argtypes = [vp, f32, f32, i64, i64, i64, vp, vp, vp, pi64, pi64, vp]
    L.hm_pairwise_candidates.argtypes = [vp, f32, f32, i64, i64, i64, vp, vp, vp, pi64, vp]
    L.hm_row_vs_all.argtypes = [vp, i64, i64, f32, vp, vp]
    L.hm_row_argmin.argtypes = [vp, i64, i64, f32, f32, pf32, pi32, pi32, pi32, vp]
    L.hm_pair_distance.argtypes = [vp, vp, vp, i64, f32, vp, vp]
    L.hm_midpoint_batch.argtypes = [vp, vp, vp, vp, i64, f32, vp, vp]
    L.hm_merge_append.argtypes = [vp, i32, i32, f32, f32, vp, i64, i64, vp]
    L.hm_batch_distance.argtypes = [vp, i64, vp, i64, i64, i64, C.c_int, f32, C.c_int, vp, vp]
    L.hm_rows_minkowski.argtypes = [vp, vp, i64, i64, C.c_int, C.c_int, vp, vp]
    L.hm_rows_distance.argtypes = [vp, vp, i64, i64, C.c_int, f32, C.c_int, vp, vp]
    L.hm_rows_log_map.argtypes = [vp, vp, i64, i64, C.c_int, C.c_int, vp, i64, vp]
    L.hm_rows_exp_map.argtypes = [vp, vp, i64, i64, C.c_int, vp, i64, vp]
    L.hm_rows_project.argtypes = [vp, i64, i64, C.c_int, f32, vp, i64, vp]
    L.hm_last_scan_stats.argtypes = [vp, pf32, pi64, pi64, pi32]
    L.hm_scan_totals.argtypes = [vp, C.POINTER(C.c_double), pi64, pi64, C.c_int]
    for name in EXPORTED_SYMBOLS:
        fn = getattr(L, name)
        if name not in ("hm_last_error", "hm_rows", "hm_tokenize_table_capacity"):
            fn.restype = C.c_int
    _lib = L
    return L


def check(status: int, engine=None) -> None:
    if status != HM_OK:
        msg = load().hm_last_error(engine)
        raise HypMergeError(status, msg.decode("utf-8", "replace") if msg else "")
