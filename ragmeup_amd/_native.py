"""ctypes binding of librmu.so (include/rmu.h).  No CPU fallback: if the library is missing or a
call fails, this raises -- the product path never routes through the oracle."""
from __future__ import annotations

import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "lib", "librmu.so")
# A/B of two builds on one GPU box (tools/enc_ab.sh lib_prev:RMU_LIB=...): honoured only with RMU_TUNING=1, like every tuning switch
if os.environ.get("RMU_TUNING") == "1" and os.environ.get("RMU_LIB"):
    SO_PATH = os.path.abspath(os.environ["RMU_LIB"])

RMU_OK = 0
METRIC_IP, METRIC_COSINE, METRIC_L2SQ = 0, 1, 2
F_Q_DEVICE, F_OUT_DEVICE, F_SMALLER_BETTER = 1, 2, 4
OPT_SCREEN = 1
OPT_SCREEN_MIN_NQ = 2
OPT_LADDER_RATIO, OPT_LADDER_FIRST = 3, 4
STAT_CAPACITY, STAT_GROW_COUNT, STAT_GROW_MS, STAT_LIVE_ROWS = 1, 2, 3, 4
COMM_ID_BYTES = 128
MAX_K = 112
MAX_DIM = 768

# every symbol include/rmu.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "rmu_init", "rmu_last_error", "rmu_version",
    "rmu_index_create", "rmu_index_free", "rmu_index_size", "rmu_index_dim", "rmu_index_metric", "rmu_index_set_option", "rmu_index_stat", "rmu_index_reserve", "rmu_index_add",
    "rmu_index_remove_rows", "rmu_index_get_rows", "rmu_index_save", "rmu_index_load", "rmu_index_mmr", "rmu_index_search_mmr", "rmu_index_search", "rmu_topk_merge",
    "rmu_last_scan_ms", "rmu_last_search_ms", "rmu_last_scan_geometry", "rmu_set_timing", "rmu_last_screened", "rmu_probe_mfma_rate",
    "rmu_comm_unique_id", "rmu_comm_init", "rmu_comm_free", "rmu_comm_world", "rmu_shard_allgather_topk", "rmu_index_screen_candidates",
    "rmu_bert_create", "rmu_bert_free", "rmu_bert_encode", "rmu_bert_encode_host", "rmu_bert_search_mmr",
    "rmu_tok_create", "rmu_tok_free", "rmu_tok_vocab_size", "rmu_tok_encode", "rmu_tok_encode_blob",
]


class RmuError(RuntimeError):
    def __init__(self, code: int, where: str, msg: str):
        super().__init__(f"{where} failed (rc={code}): {msg}")
        self.code = code


_lib = None
_lock = threading.Lock()


def _declare(lib):
    c = ctypes
    vp, i64, i32, u32, u64, f32 = c.c_void_p, c.c_int64, c.c_int, c.c_uint, c.c_uint64, c.c_float
    lib.rmu_init.argtypes = [i32]
    lib.rmu_last_error.restype = c.c_char_p
    lib.rmu_version.restype = c.c_char_p
    lib.rmu_index_create.argtypes = [c.POINTER(vp), i32, i32, i64]
    lib.rmu_index_free.argtypes = [vp]
    lib.rmu_index_size.argtypes = [vp, c.POINTER(i64)]
    lib.rmu_index_dim.argtypes = [vp, c.POINTER(i32)]
    lib.rmu_index_metric.argtypes = [vp, c.POINTER(i32)]
    lib.rmu_index_set_option.argtypes = [vp, i32, i64]
    lib.rmu_index_stat.argtypes = [vp, i32, c.POINTER(c.c_double)]
    lib.rmu_index_reserve.argtypes = [vp, i64]
    lib.rmu_comm_unique_id.argtypes = [vp]
    lib.rmu_comm_init.argtypes = [c.POINTER(vp), vp, i32, i32]
    lib.rmu_comm_free.argtypes = [vp]
    lib.rmu_comm_world.argtypes = [vp, c.POINTER(i32), c.POINTER(i32)]
    lib.rmu_shard_allgather_topk.argtypes = [vp, vp, vp, i64, i32, u32, vp, vp, u64]
    lib.rmu_index_screen_candidates.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    lib.rmu_index_add.argtypes = [vp, vp, i64, i32, c.POINTER(i64)]
    lib.rmu_index_remove_rows.argtypes = [vp, vp, i64, c.POINTER(i64)]
    lib.rmu_index_get_rows.argtypes = [vp, vp, i64, vp]
    lib.rmu_index_mmr.argtypes = [vp, vp, i64, vp, i32, i32, c.c_double, u32, vp]
    lib.rmu_index_search_mmr.argtypes = [vp, vp, i64, i32, i32, c.c_double, i64, vp, vp]
    lib.rmu_index_save.argtypes = [vp, c.c_char_p]
    lib.rmu_index_load.argtypes = [c.POINTER(vp), c.c_char_p]
    lib.rmu_index_search.argtypes = [vp, vp, i64, i32, u32, i64, vp, vp, u64]
    lib.rmu_topk_merge.argtypes = [vp, vp, i32, i64, i32, u32, vp, vp, u64]
    lib.rmu_last_scan_ms.restype = f32
    lib.rmu_last_search_ms.restype = f32
    lib.rmu_last_scan_geometry.argtypes = [c.POINTER(i32)] * 4
    lib.rmu_set_timing.argtypes = [i32]
    lib.rmu_probe_mfma_rate.argtypes = [i32, i32, i32, c.POINTER(c.c_double)]
    lib.rmu_tok_create.argtypes = [c.POINTER(vp), c.c_char_p, i32]
    lib.rmu_tok_free.argtypes = [vp]
    lib.rmu_tok_vocab_size.argtypes = [vp]
    lib.rmu_tok_encode.argtypes = [vp, c.POINTER(c.c_char_p), c.POINTER(c.c_char_p), i32, i32, vp, vp, vp]
    lib.rmu_tok_encode_blob.argtypes = [vp, c.c_char_p, i64, c.c_char_p, i64, i32, i32, vp, vp, vp]
    if hasattr(lib, "rmu_bert_create"):
        lib.rmu_bert_create.argtypes = [c.POINTER(vp), vp, c.POINTER(vp), i32]
        lib.rmu_bert_free.argtypes = [vp]
        lib.rmu_bert_encode.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, i64, u64]
        lib.rmu_bert_encode_host.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, i64]
        lib.rmu_bert_search_mmr.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, c.c_double, i64, vp, vp, vp]


def lib():
    """Load librmu.so.  Import torch FIRST when torch is used in the same process: both link
    libamdhip64.so.7 and the already-loaded (torch-bundled) runtime is then shared."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(SO_PATH):
                    raise RuntimeError(
                        f"{SO_PATH} is missing: build it with `python -m ragmeup_amd.build` "
                        "(there is no CPU fallback for the MI355X path)")
                l = ctypes.CDLL(SO_PATH, mode=ctypes.RTLD_GLOBAL)
                _declare(l)
                _lib = l
    return _lib


def check(rc: int, where: str):
    if rc != RMU_OK:
        raise RmuError(rc, where, (lib().rmu_last_error() or b"").decode("utf-8", "replace"))
