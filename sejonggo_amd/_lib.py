"""ctypes loader for libsgo_hip.so (C ABI: include/sgo.h).  Fails loudly when the library or a HIP
device is missing -- there is no CPU fallback in the product path."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsgo_hip.so")

SUPPORTED_SIZES = (5, 7, 9, 13, 19)

ABI_VERSION = 3          # SGO_ABI_VERSION of include/sgo.h these bindings were written against
SGO_OK = 0
SGO_ERR_OCCUPIED = -101
SGO_ERR_RANGE = -102
SGO_ERR_UNSUPPORTED = -3

# every symbol include/sgo.h declares (tests/test_abi.py checks the built library exports them all)
SYMBOLS = [
    "sgo_last_error", "sgo_version", "sgo_device_count", "sgo_set_device", "sgo_plane_words", "sgo_packed_words",
    "sgo_apad", "sgo_game_init", "sgo_make_play", "sgo_take_stones", "sgo_board_query", "sgo_legal_moves", "sgo_get_winner", "sgo_sym_apply",
    "sgo_sym_invert_policy", "sgo_sym_lut", "sgo_pack_dev", "sgo_unpack_dev", "sgo_advance_legal_dev",
    "sgo_legal_dev", "sgo_score_dev", "sgo_nn_pack_dev", "sgo_bias_act_dev", "sgo_conv3x3_bias_act_dev", "sgo_conv3x3_tower_dev", "sgo_conv3x3_tower_packed_bytes", "sgo_conv3x3_tower_prepack_dev", "sgo_conv3x3_tower_packed_dev", "sgo_conv_packed_variant", "sgo_conv3x3_stem_dev", "sgo_conv_tile_order", "sgo_conv_tower_kernel", "sgo_conv_tower_slice_cap", "sgo_advance_mode", "sgo_debug_counters", "sgo_ctx_create", "sgo_ctx_destroy", "sgo_blocks_per_game", "sgo_pool_info", "sgo_start_games", "sgo_start_games2", "sgo_eval_models",
    "sgo_step", "sgo_step_enqueue", "sgo_step_status", "sgo_eval_list", "sgo_stem_packed_dev", "sgo_collect", "sgo_drain_records", "sgo_game_results", "sgo_root_table", "sgo_tree_serialize", "sgo_tree_dump",
    "sgo_game_board", "sgo_set_halt", "sgo_advance_timing",
]


class SgoError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("size", C.c_int32), ("n_games", C.c_int32), ("sims", C.c_int32), ("energy", C.c_int32),
                ("stop_exploration", C.c_int32), ("num_moves", C.c_int32), ("blocks_per_game", C.c_int32),
                ("self_play", C.c_int32), ("komi", C.c_double), ("dirichlet_epsilon", C.c_double),
                ("device_id", C.c_int32), ("two_model", C.c_int32), ("shared_blocks", C.c_int32), ("reserved", C.c_int32)]


class Status(C.Structure):
    _fields_ = [("n_eval", C.c_int32), ("n_records", C.c_int32), ("n_active", C.c_int32), ("n_done", C.c_int32),
                ("error", C.c_int32), ("error_game", C.c_int32), ("total_moves", C.c_int64),
                ("total_evals", C.c_int64), ("none_events", C.c_int64)]


class MoveRecord(C.Structure):
    _fields_ = [("game", C.c_int32), ("game_seq", C.c_int32), ("move_n", C.c_int32), ("action", C.c_int32),
                ("player", C.c_int32), ("value", C.c_float)]


class GameResult(C.Structure):
    _fields_ = [("winner", C.c_int32), ("black", C.c_int32), ("white", C.c_double), ("end_reason", C.c_int32),
                ("n_moves", C.c_int32), ("last_player", C.c_int32), ("done", C.c_int32), ("first_model", C.c_int32),
                ("blocks_high_water", C.c_int32)]


MOVE_RECORD_DTYPE = np.dtype([("game", "<i4"), ("game_seq", "<i4"), ("move_n", "<i4"), ("action", "<i4"),
                              ("player", "<i4"), ("value", "<f4")])
GAME_RESULT_DTYPE = np.dtype([("winner", "<i4"), ("black", "<i4"), ("white", "<f8"), ("end_reason", "<i4"),
                              ("n_moves", "<i4"), ("last_player", "<i4"), ("done", "<i4"), ("first_model", "<i4"),
                              ("blocks_high_water", "<i4")], align=True)

_lib = None
gpu_touched_pid = None   # pid of the process in which this module first made a HIP call (fork-safety checks)


def load():
    """Returns the loaded library; raises SgoError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SgoError("libsgo_hip.so is not built (run `python -m sejonggo_amd.build` or __graft_entry__.build()); "
                       "the MI355X path has no CPU fallback")
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7; two HIP runtimes in one process cannot both
    # own the GPU.  Importing torch first makes the dynamic linker resolve our NEEDED libamdhip64.so.7 to
    # the copy torch already loaded, so the engine and the net share one runtime (streams, pointers).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    if lib.sgo_version() != ABI_VERSION:
        raise SgoError("libsgo_hip.so speaks ABI version %d, these bindings version %d (include/sgo.h SGO_ABI_VERSION): "
                       "rebuild with `python -m sejonggo_amd.build`" % (lib.sgo_version(), ABI_VERSION))
    lib.sgo_last_error.restype = C.c_char_p
    lib.sgo_ctx_create.restype = C.c_void_p
    lib.sgo_ctx_create.argtypes = [C.POINTER(Config)]
    lib.sgo_ctx_destroy.restype = None
    lib.sgo_ctx_destroy.argtypes = [C.c_void_p]
    lib.sgo_tree_serialize.restype = C.c_int64
    lib.sgo_tree_dump.restype = C.c_int64
    lib.sgo_conv3x3_bias_act_dev.argtypes = [C.c_int] * 6 + [C.c_void_p] * 6
    lib.sgo_conv3x3_tower_dev.argtypes = [C.c_int] * 3 + [C.c_void_p] * 6
    lib.sgo_conv3x3_tower_packed_dev.argtypes = [C.c_int] * 3 + [C.c_void_p] * 6
    lib.sgo_conv3x3_tower_prepack_dev.argtypes = [C.c_void_p] * 3
    lib.sgo_conv3x3_tower_packed_bytes.restype = C.c_long
    lib.sgo_conv_packed_variant.argtypes = [C.c_int]
    lib.sgo_conv3x3_stem_dev.argtypes = [C.c_int] * 3 + [C.c_void_p] * 5
    lib.sgo_stem_packed_dev.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
    lib.sgo_step_enqueue.argtypes = [C.c_void_p] * 5
    lib.sgo_step_status.argtypes = [C.c_void_p, C.c_void_p]
    lib.sgo_eval_list.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.sgo_blocks_per_game.argtypes = [C.c_void_p]
    lib.sgo_pool_info.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.sgo_conv_tile_order.argtypes = [C.c_int]
    lib.sgo_conv_tower_kernel.argtypes = [C.c_int]
    lib.sgo_conv_tower_slice_cap.argtypes = [C.c_long]
    lib.sgo_conv_tower_slice_cap.restype = C.c_long
    lib.sgo_advance_mode.argtypes = [C.c_int]
    lib.sgo_bias_act_dev.argtypes = [C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = lib
    return lib


def check(rc, what=""):
    if rc < 0:
        raise SgoError("%s failed (%d): %s" % (what, rc, load().sgo_last_error().decode()))
    return rc


def gpu_runtime_initialised():
    """True when THIS process has initialised a HIP runtime (through this library or through torch.cuda): a child
    forked from it cannot use the GPU; start such children with the 'spawn' method instead."""
    import sys
    if gpu_touched_pid == os.getpid():
        return True
    torch = sys.modules.get("torch")
    return bool(torch is not None and torch.cuda.is_initialized())


def require_gpu():
    global gpu_touched_pid
    lib = load()
    if gpu_touched_pid is not None and gpu_touched_pid != os.getpid():
        raise SgoError("this process was forked from one that had already initialised the GPU runtime; HIP does not "
                       "survive a fork.  Start GPU workers before touching the GPU, or with the 'spawn' start method "
                       "(sejonggo_amd.selfplay_worker does this by itself)")
    gpu_touched_pid = os.getpid()
    if lib.sgo_device_count() <= 0:
        raise SgoError("no HIP device visible: the sejonggo_amd hot path runs on MI355X only (no CPU fallback)")
    return lib


def ptr(a):
    """void* of a numpy array or a torch tensor (host or device), or None."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(a.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
