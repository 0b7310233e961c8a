"""ctypes binding of libg1s_diff.so (include/g1s_diff.h).

The shared library is built in-tree by `__graft_entry__.build()` (or
`make -C grav1synth_amd/csrc`).  There is NO fallback: if the library is
missing, importing the compute entry points raises, loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (GPU_MAX_HW_QUEUES=8 -- a hardware queue per stream, INTEGRATION.md section 3 -- is the PROCESS' choice: the entry points
#  set it (bench.py, the command in cli.py, tests/conftest.py), importing this package changes nobody's environment; the
#  engine says so once on stderr when a generator's streams are made with fewer queues than they want)
LIB_PATH = os.environ.get("G1S_LIB") or os.path.join(_HERE, "libg1s_diff.so")  # (G1S_LIB: an instrumented build, tools/ only)

G1S_OK = 0
G1S_ERR_CAPACITY = -8
ERRORS = {
    -1: "G1S_ERR_INVALID",
    -2: "G1S_ERR_DIM_MISMATCH",
    -3: "G1S_ERR_NOT_ENOUGH_FLAT",
    -4: "G1S_ERR_SOLVE",
    -5: "G1S_ERR_NO_DEVICE",
    -6: "G1S_ERR_HIP",
    -7: "G1S_ERR_STATE",
    -8: "G1S_ERR_CAPACITY",
    -9: "G1S_ERR_UNSUPPORTED",
}


class G1SFrame(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("bytes_per_sample", C.c_uint8),
        ("xdec", C.c_uint8),
        ("ydec", C.c_uint8),
        ("nplanes", C.c_uint8),
        ("data", C.c_void_p * 3),
        ("stride_bytes", C.c_size_t * 3),
        ("on_device", C.c_int32),
    ]


class G1SSegment(C.Structure):
    _fields_ = [
        ("start_time", C.c_uint64),
        ("end_time", C.c_uint64),
        ("random_seed", C.c_uint16),
        ("num_y_points", C.c_uint8),
        ("num_cb_points", C.c_uint8),
        ("num_cr_points", C.c_uint8),
        ("scaling_points_y", (C.c_uint8 * 2) * 14),
        ("scaling_points_cb", (C.c_uint8 * 2) * 10),
        ("scaling_points_cr", (C.c_uint8 * 2) * 10),
        ("scaling_shift", C.c_uint8),
        ("ar_coeff_lag", C.c_uint8),
        ("num_y_coeffs", C.c_uint8),
        ("num_uv_coeffs", C.c_uint8),
        ("ar_coeffs_y", C.c_int8 * 24),
        ("ar_coeffs_cb", C.c_int8 * 25),
        ("ar_coeffs_cr", C.c_int8 * 25),
        ("ar_coeff_shift", C.c_uint8),
        ("cb_mult", C.c_uint8),
        ("cb_luma_mult", C.c_uint8),
        ("cb_offset", C.c_uint16),
        ("cr_mult", C.c_uint8),
        ("cr_luma_mult", C.c_uint8),
        ("cr_offset", C.c_uint16),
        ("chroma_scaling_from_luma", C.c_uint8),
        ("grain_scale_shift", C.c_uint8),
        ("overlap_flag", C.c_uint8),
    ]


class G1SOpts(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device", C.c_int32),
        ("ar_coeff_lag", C.c_uint32),
        ("luma_only", C.c_uint32),
        ("batch_frames", C.c_uint32),
        ("records_only", C.c_uint32),
    ]


class G1SStats(C.Structure):
    _fields_ = [
        ("frames", C.c_uint64),
        ("blocks", C.c_uint64),
        ("flat_blocks", C.c_uint64),
        ("ms_flat_features", C.c_double),
        ("ms_flat_select", C.c_double),
        ("ms_ar_accumulate", C.c_double),
        ("ms_total_gpu", C.c_double),
        ("launches_flat_features", C.c_uint64),
        ("launches_flat_select", C.c_uint64),
        ("launches_ar_accumulate", C.c_uint64),
        ("ms_host_fold", C.c_double),
        ("ms_residual", C.c_double),
        ("literal_blocks", C.c_uint64),
        ("ms_chain", C.c_double),
        ("chain_batches", C.c_uint64),
    ]


class G1SY4MInfo(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("bit_depth", C.c_uint32),
        ("xdec", C.c_uint32),
        ("ydec", C.c_uint32),
        ("nplanes", C.c_uint32),
        ("fps_num", C.c_int64),
        ("fps_den", C.c_int64),
    ]


class G1SFilterDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("top", C.c_uint64),
        ("bottom", C.c_uint64),
        ("left", C.c_uint64),
        ("right", C.c_uint64),
        ("width", C.c_uint64),
        ("height", C.c_uint64),
        ("alg", C.c_char * 16),
    ]


NEXT_FRAME_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(G1SFrame))

# every symbol include/g1s_diff.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("g1s_diff_new", C.c_void_p, [C.c_int64, C.c_int64, C.c_uint32, C.c_uint32, C.POINTER(G1SOpts)]),
    ("g1s_last_global_error", C.c_char_p, []),
    ("g1s_diff_frame", C.c_int, [C.c_void_p, C.POINTER(G1SFrame), C.POINTER(G1SFrame)]),
    ("g1s_diff_frames", C.c_int, [C.c_void_p, C.POINTER(G1SFrame), C.POINTER(G1SFrame), C.c_size_t]),
    ("g1s_diff_sync", C.c_int, [C.c_void_p]),
    ("g1s_diff_finish", C.c_int, [C.c_void_p, C.POINTER(G1SSegment), C.c_size_t, C.POINTER(C.c_size_t)]),
    ("g1s_diff_free", None, [C.c_void_p]),
    ("g1s_diff_last_error", C.c_char_p, [C.c_void_p]),
    ("g1s_record_size", C.c_size_t, [C.c_uint32] * 6),
    ("g1s_record_init", C.c_int, [C.c_void_p, C.c_size_t] + [C.c_uint32] * 6),
    ("g1s_diff_take_records", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("g1s_fold_new", C.c_void_p, [C.c_int64, C.c_int64, C.c_uint32]),
    ("g1s_fold_push", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("g1s_diff_take_latest", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("g1s_latest_size", C.c_size_t, [C.c_uint32]),
    ("g1s_shard_msg_size", C.c_size_t, [C.c_uint32, C.c_uint32]),
    ("g1s_shard_pack", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    ("g1s_shard_msg_from_latest", C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]),
    ("g1s_shard_msg_from_latest_at", C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_size_t]),
    ("g1s_shard_merge", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32]),
    ("g1s_latest_from_record", C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t]),
    ("g1s_latest_from_records", C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t]),
    ("g1s_usable_cpus", C.c_uint, []),
    ("g1s_shard_flush_rounds", C.c_uint, []),
    ("g1s_fold_push_latest", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]),
    ("g1s_fold_push_many", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]),
    ("g1s_fold_finish", C.c_int, [C.c_void_p, C.POINTER(G1SSegment), C.c_size_t, C.POINTER(C.c_size_t)]),
    ("g1s_fold_free", None, [C.c_void_p]),
    ("g1s_fold_last_error", C.c_char_p, [C.c_void_p]),
    ("g1s_format_tbl", C.c_long, [C.POINTER(G1SSegment), C.c_size_t, C.c_char_p, C.c_size_t]),
    ("g1s_write_tbl", C.c_int, [C.c_char_p, C.POINTER(G1SSegment), C.c_size_t]),
    ("g1s_parse_tbl", C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(G1SSegment), C.c_size_t, C.POINTER(C.c_size_t),
                                C.c_char_p, C.c_size_t]),
    ("g1s_tbl_segment_for", C.c_long, [C.POINTER(G1SSegment), C.c_size_t, C.c_uint64]),
    ("g1s_diff_get_stats", C.c_int, [C.c_void_p, C.POINTER(G1SStats)]),
    ("g1s_diff_set_timing", C.c_int, [C.c_void_p, C.c_int]),
    ("g1s_diff_kernel_times", C.c_long, [C.c_void_p, C.c_char_p, C.c_size_t]),
    ("g1s_diff_frames_released", C.c_uint64, [C.c_void_p]),
    ("g1s_diff_frames_copied", C.c_uint64, [C.c_void_p, C.c_uint64]),
    ("g1s_diff_set_flat_finder", C.c_int, [C.c_void_p, C.c_int]),
    ("g1s_diff_last_record", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("g1s_record_geometry", C.c_int, [C.c_void_p] + [C.POINTER(C.c_uint32)] * 4),
    ("g1s_record_flat_mask", C.POINTER(C.c_uint8), [C.c_void_p]),
    ("g1s_record_scores", C.POINTER(C.c_float), [C.c_void_p]),
    ("g1s_record_ar_sums", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.POINTER(C.c_int64)),
                                      C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.c_int64)]),
    ("g1s_record_block_stats", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.POINTER(C.c_uint32)),
                                          C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_uint32))]),
    ("g1s_diff_run", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    ("g1s_filters_new", C.c_void_p, [C.c_char_p, C.c_char_p, C.c_size_t]),
    ("g1s_filters_len", C.c_size_t, [C.c_void_p]),
    ("g1s_filters_get", C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(G1SFilterDesc)]),
    ("g1s_filters_apply", C.c_int, [C.c_void_p, C.POINTER(G1SFrame), C.POINTER(G1SFrame), C.c_char_p, C.c_size_t]),
    ("g1s_filters_apply_bd", C.c_int, [C.c_void_p, C.POINTER(G1SFrame), C.c_uint32, C.c_int32, C.c_uint32, C.POINTER(G1SFrame), C.c_char_p,
                                        C.c_size_t]),
    ("g1s_filters_has_resize", C.c_int, [C.c_void_p]),
    ("g1s_fold_frames", C.c_uint64, [C.c_void_p]),
    ("g1s_resize_plan", C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p, C.c_size_t]),
    ("g1s_resize_frame_to_host", C.c_int, [C.c_char_p, C.POINTER(G1SFrame), C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32,
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]),
    ("g1s_filters_free", None, [C.c_void_p]),
    ("g1s_diff_run_filtered", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    ("g1s_estimate_new", C.c_void_p, [C.c_uint32, C.c_int32, C.c_uint32]),
    ("g1s_estimate_frame", C.c_int, [C.c_void_p, C.POINTER(G1SFrame)]),
    ("g1s_estimate_finish", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.POINTER(C.c_size_t)]),
    ("g1s_estimate_set_timing", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    ("g1s_estimate_last_error", C.c_char_p, [C.c_void_p]),
    ("g1s_estimate_free", None, [C.c_void_p]),
    ("g1s_format_estimates", C.c_long, [C.POINTER(C.c_double), C.c_size_t, C.c_char_p, C.c_size_t]),
    ("g1s_y4m_open", C.c_void_p, [C.c_char_p, C.c_char_p, C.c_size_t]),
    ("g1s_y4m_get_info", C.c_int, [C.c_void_p, C.POINTER(G1SY4MInfo)]),
    ("g1s_y4m_next", C.c_int, [C.c_void_p, C.POINTER(G1SFrame)]),
    ("g1s_y4m_bind", C.c_int, [C.c_void_p, C.c_void_p]),
    ("g1s_y4m_last_error", C.c_char_p, [C.c_void_p]),
    ("g1s_y4m_close", None, [C.c_void_p]),
    ("g1s_diff_y4m_files", C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(G1SOpts),
                                     C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    ("g1s_diff_y4m_files_filtered", C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(G1SOpts), C.c_char_p,
                                              C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    ("g1s_diff_y4m_files_sharded", C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(G1SOpts), C.c_char_p,
                                              C.POINTER(C.c_int32), C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_int),
                                              C.c_char_p, C.c_size_t]),
]

_lib = None


def lib() -> C.CDLL:
    """Load libg1s_diff.so; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C grav1synth_amd/csrc). "
            "grav1synth_amd has no CPU fallback for the diff path."
        )
    # PyTorch-ROCm ships its own libamdhip64: it must be the HIP runtime of the process (device pointers of
    # torch tensors are handed to the kernels), so it has to be mapped before libg1s_diff.so asks the
    # dynamic linker for that soname -- two runtimes in one process do not see each other's device state.
    try:
        import torch  # noqa: F401
    except ImportError:  # host-only use (fold, .tbl writer): the system HIP runtime will do
        pass
    L = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(L, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


class G1SError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{ERRORS.get(code, code)}: {message}")
        self.code = code
        self.message = message
