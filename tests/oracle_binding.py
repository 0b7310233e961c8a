"""ctypes binding of the CPU oracle (oracle/liborc_diff.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never imported by the grav1synth_amd package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(ORACLE_DIR, "liborc_diff.so")


class OrcFrame(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("bytes_per_sample", C.c_uint8),
        ("xdec", C.c_uint8),
        ("ydec", C.c_uint8),
        ("nplanes", C.c_uint8),
        ("data", C.c_void_p * 3),
        ("stride_bytes", C.c_size_t * 3),
    ]


class OrcSegment(C.Structure):
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


_lib = None


def build() -> None:
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def _prototypes(L):
    L.orc_estimate_plane_noise.restype = C.c_double
    L.orc_estimate_plane_noise.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32]
    L.orc_resize_plan.restype = C.c_int
    L.orc_resize_plan.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    L.orc_resize_plane.restype = C.c_int
    L.orc_resize_plane.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                   C.c_int]
    L.orc_diff_new.restype = C.c_void_p
    L.orc_diff_new.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_diff_frame.restype = C.c_int
    L.orc_diff_frame.argtypes = [C.c_void_p, C.POINTER(OrcFrame), C.POINTER(OrcFrame)]
    L.orc_diff_finish.restype = C.c_int
    L.orc_diff_finish.argtypes = [C.c_void_p, C.POINTER(OrcSegment), C.c_int]
    L.orc_diff_free.argtypes = [C.c_void_p]
    L.orc_diff_last_error.restype = C.c_char_p
    L.orc_diff_last_error.argtypes = [C.c_void_p]
    L.orc_last_flat_mask.restype = C.POINTER(C.c_uint8)
    L.orc_last_flat_mask.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_last_scores.restype = C.POINTER(C.c_float)
    L.orc_last_scores.argtypes = [C.c_void_p]
    L.orc_last_ar_sums.restype = C.c_int
    L.orc_last_ar_sums.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    L.orc_last_block_stats.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_num_segments.restype = C.c_int
    L.orc_num_segments.argtypes = [C.c_void_p]
    L.orc_format_tbl.restype = C.c_long
    L.orc_format_tbl.argtypes = [C.POINTER(OrcSegment), C.c_int, C.c_char_p, C.c_size_t]
    if hasattr(L, "orc_diff_save"):
        L.orc_diff_state_size.restype = C.c_size_t
        L.orc_diff_state_size.argtypes = [C.c_void_p]
        L.orc_diff_save.restype = C.c_long
        L.orc_diff_save.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_diff_restore.restype = C.c_int
        L.orc_diff_restore.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    return L


def load_variant(name: str):
    """A pin-sensitivity build of the oracle (oracle/Makefile `variants`; tools/pin_sensitivity.py): the same restatement with fused
    multiply-adds at a group of sites / contraction left to the compiler / one division of the exact sums a frame."""
    path = os.path.join(ORACLE_DIR, "_variants", f"liborc_{name}.so")
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, f"_variants/liborc_{name}.so"])
    return _prototypes(C.CDLL(path))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    if not hasattr(L, "orc_estimate_plane_noise") or not hasattr(L, "orc_resize_plane") or not hasattr(L, "orc_diff_save"):  # a library from before these were added
        del L
        build()
        L = C.CDLL(_LIB_PATH)
    _lib = _prototypes(L)
    return _lib


def estimate_plane_noise(plane: np.ndarray, bit_depth: int) -> Optional[float]:
    """av1_grain::estimate_plane_noise as restated in oracle/estimate_oracle.c (None = fewer than 16 smooth pixels)."""
    p = np.ascontiguousarray(plane)
    assert p.dtype == (np.uint8 if bit_depth == 8 else np.uint16)
    v = lib().orc_estimate_plane_noise(p.ctypes.data, p.strides[0], p.shape[1], p.shape[0], bit_depth)
    return None if v == -1.0 else float(v)


def resize_plan(alg: str, src: int, dst: int):
    """taps of one axis as oracle/resize_oracle.c forms them: (idx[dst, taps], coef[dst, taps])"""
    L = lib()
    taps = L.orc_resize_plan(alg.encode(), src, dst, None, None, 0)
    assert taps > 0
    idx = np.zeros((dst, taps), np.int32)
    coef = np.zeros((dst, taps), np.float32)
    assert L.orc_resize_plan(alg.encode(), src, dst, idx.ctypes.data, coef.ctypes.data, idx.size) == taps
    return idx, coef


def resize_planes(planes: Sequence[np.ndarray], xdec: int, ydec: int, width: int, height: int, bit_depth: int, alg: str = "catmullrom"):
    """video_resize::resize as restated in oracle/resize_oracle.c: every plane, chroma to (width >> xdec, height >> ydec)."""
    L = lib()
    out = []
    for c, p in enumerate(planes):
        p = np.ascontiguousarray(p)
        dw, dh = (width >> xdec, height >> ydec) if c else (width, height)
        o = np.zeros((dh, dw), p.dtype)
        rc = L.orc_resize_plane(alg.encode(), p.ctypes.data, p.dtype.itemsize, p.strides[0], p.shape[1], p.shape[0], o.ctypes.data, o.strides[0],
                                dw, dh, bit_depth)
        assert rc == 0
        out.append(o)
    return out


def _np_frame(planes: Sequence[np.ndarray], xdec: int, ydec: int) -> OrcFrame:
    f = OrcFrame()
    f.width = planes[0].shape[1]
    f.height = planes[0].shape[0]
    f.bytes_per_sample = planes[0].dtype.itemsize
    f.xdec, f.ydec = xdec, ydec
    f.nplanes = len(planes)
    for i, p in enumerate(planes):
        assert p.flags["C_CONTIGUOUS"] or p.strides[1] == p.dtype.itemsize
        f.data[i] = p.ctypes.data
        f.stride_bytes[i] = p.strides[0]
    return f


class OracleDiff:
    """The oracle behind the same three-method shape as av1_grain::DiffGenerator
    (reference src/main.rs:420-427, :442, :524)."""

    def __init__(self, fps_num: int, fps_den: int, src_bd: int, den_bd: int, lag: int = 3, chroma: bool = True, library=None):
        self.L = library if library is not None else lib()  # (library: a pin-sensitivity variant, load_variant)
        self.h = self.L.orc_diff_new(fps_num, fps_den, src_bd, den_bd, lag, int(chroma))
        if not self.h:
            raise ValueError("orc_diff_new failed (lag must be 1..3)")
        self.lag = lag

    def diff_frame(self, src: Sequence[np.ndarray], den: Sequence[np.ndarray], xdec: int = 1, ydec: int = 1) -> None:
        fs = _np_frame(src, xdec, ydec)
        fd = _np_frame(den, xdec, ydec)
        rc = self.L.orc_diff_frame(self.h, C.byref(fs), C.byref(fd))
        if rc != 0:
            raise RuntimeError(self.L.orc_diff_last_error(self.h).decode())

    def finish(self) -> List[OrcSegment]:
        arr = (OrcSegment * 256)()
        n = self.L.orc_diff_finish(self.h, arr, 256)
        if n < 0:
            raise RuntimeError("orc_diff_finish failed")
        return [arr[i] for i in range(n)]

    # ---- last-frame introspection ----
    def flat_mask(self) -> np.ndarray:
        nbw, nbh = C.c_int(), C.c_int()
        p = self.L.orc_last_flat_mask(self.h, C.byref(nbw), C.byref(nbh))
        return np.ctypeslib.as_array(p, shape=(nbh.value, nbw.value)).copy()

    def scores(self) -> np.ndarray:
        m = self.flat_mask()
        p = self.L.orc_last_scores(self.h)
        return np.ctypeslib.as_array(p, shape=m.shape).copy()

    def ar_sums(self, c: int):
        n = (2 * self.lag + 1) ** 2 // 2 + (1 if c else 0)
        S = np.zeros((n, n), dtype=np.int64)
        Sb = np.zeros(n, dtype=np.int64)
        nobs = C.c_int64()
        self.L.orc_last_ar_sums(self.h, c, S.ctypes.data, Sb.ctypes.data, C.byref(nobs))
        return S, Sb, nobs.value

    def block_stats(self, c: int):
        m = self.flat_mask()
        nb = m.size
        ls = np.zeros(nb, dtype=np.uint32)
        sd = np.zeros(nb, dtype=np.int32)
        sd2 = np.zeros(nb, dtype=np.uint32)
        self.L.orc_last_block_stats(self.h, c, ls.ctypes.data, sd.ctypes.data, sd2.ctypes.data)
        return ls, sd, sd2

    def num_segments(self) -> int:
        return self.L.orc_num_segments(self.h)

    def save(self) -> bytes:
        """the state that crosses frames (checkpoint of a long job)"""
        n = self.L.orc_diff_state_size(self.h)
        buf = C.create_string_buffer(n)
        assert self.L.orc_diff_save(self.h, buf, n) == n
        return buf.raw

    def restore(self, state: bytes) -> None:
        rc = self.L.orc_diff_restore(self.h, state, len(state))
        if rc:
            raise ValueError(f"orc_diff_restore failed ({rc})")

    def close(self):
        if self.h:
            self.L.orc_diff_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def format_tbl(segs: Sequence[OrcSegment]) -> bytes:
    L = lib()
    arr = (OrcSegment * len(segs))(*segs)
    buf = C.create_string_buffer(1 << 20)
    n = L.orc_format_tbl(arr, len(segs), buf, len(buf))
    if n < 0:
        raise RuntimeError("orc_format_tbl overflow")
    return buf.raw[:n]
