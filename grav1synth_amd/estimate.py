"""`grav1synth estimate` on an MI355X: av1_grain::estimate_plane_noise per frame (N4; /root/reference/src/main.rs:534-608).

>>> est = NoiseEstimator(10)
>>> est.estimate_frame(y_plane)                 # numpy / torch (host or cuda), u8 or u16; only the luma plane is read
>>> estimates = est.finish()                    # List[Optional[float]]: None where the reference has None
>>> open(out, "wb").write(format_estimates(estimates))     # "filmgrn1" + one "{:.3}" line per frame (-1 for None)
"""
from __future__ import annotations

import ctypes as C
import logging
from typing import List, Optional

from . import _lib
from ._lib import G1SError
from .diff import Frame

log = logging.getLogger("grav1synth")


class NoiseEstimator:
    def __init__(self, bit_depth: int, *, device: int = -1, batch_frames: int = 0):
        self._L = _lib.lib()
        self._h = self._L.g1s_estimate_new(bit_depth, device, batch_frames)
        if not self._h:
            raise G1SError(-5, "g1s_estimate_new failed: no HIP device (the estimator has no CPU fallback) or bit depth outside 8..16")
        self._keep: list = []

    def _check(self, rc: int) -> None:
        if rc:
            raise G1SError(rc, self._L.g1s_estimate_last_error(self._h).decode())

    def estimate_frame(self, y_plane) -> None:
        keep: list = []
        f = Frame([y_plane], 1, 1).to_c(keep)
        if f.on_device == 1:
            import torch
            torch.cuda.current_stream().synchronize()
            self._keep.extend(keep)
        elif f.on_device == 2:
            f.on_device = 0  # (pinned or not: this entry point copies before it returns)
        self._check(self._L.g1s_estimate_frame(self._h, C.byref(f)))

    def finish(self) -> List[Optional[float]]:
        n = C.c_size_t()
        cap = 4096
        buf = (C.c_double * cap)()
        rc = self._L.g1s_estimate_finish(self._h, buf, cap, C.byref(n))
        if rc == _lib.G1S_ERR_CAPACITY:
            cap = n.value
            buf = (C.c_double * cap)()
            rc = self._L.g1s_estimate_finish(self._h, buf, cap, C.byref(n))
        self._check(rc)
        self._keep.clear()
        return [None if buf[i] == -1.0 else float(buf[i]) for i in range(n.value)]

    def kernel_time(self, enable: bool = True):
        """(milliseconds, frames) of the kernel launches so far (HIP events); enables / disables the timing."""
        ms, fr = C.c_double(), C.c_uint64()
        self._L.g1s_estimate_set_timing(self._h, int(enable), C.byref(ms), C.byref(fr))
        return ms.value, fr.value

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.g1s_estimate_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def format_estimates(estimates) -> bytes:
    """The command's output file (src/main.rs:596-603)."""
    L = _lib.lib()
    n = len(estimates)
    arr = (C.c_double * max(n, 1))(*[-1.0 if e is None else float(e) for e in estimates])
    buf = C.create_string_buffer(16 + 32 * n)
    w = L.g1s_format_estimates(arr, n, buf, len(buf))
    if w < 0:
        raise G1SError(int(w), "g1s_format_estimates failed")
    return buf.raw[:w]


def estimate_y4m_file(source: str, output: str, *, device: int = -1) -> int:
    """`grav1synth estimate SOURCE -o OUTPUT` for a .y4m input.  Returns the number of frames."""
    from .ingest import Y4MReader

    rd = Y4MReader(source)
    est = NoiseEstimator(rd.details.bit_depth, device=device)
    n = 0
    while True:
        planes = rd.get_frame()
        if planes is None:
            break
        est.estimate_frame(planes[0])
        n += 1
    with open(output, "wb") as f:
        f.write(format_estimates(est.finish()))
    est.close()
    rd.close()
    return n
