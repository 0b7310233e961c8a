"""Caller side of the diff path: `grav1synth diff SOURCE DENOISED -o OUT` over raw-video files.

Mirrors the reference's command (src/main.rs:414-531): frame rate from the source reader, bit
depths from each reader, the frame-pair loop with its unequal-frame-count warning, finish, the
"filmgrn1" table.  The frame source is the library's YUV4MPEG2 reader (pinned, read-ahead); all
of it is native code behind the C ABI -- this module only binds it.
"""
from __future__ import annotations

import ctypes as C
import logging
from dataclasses import dataclass
from fractions import Fraction
from typing import Iterable, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import G1SFrame, G1SOpts, G1SY4MInfo

log = logging.getLogger("grav1synth_amd")

UNEQUAL_WARNING = "Videos did not have equal frame counts. Resulting grain table may not be as expected."


@dataclass
class VideoDetails:
    """What BitstreamReader::get_video_details returns (src/reader.rs:20-27)."""
    width: int
    height: int
    bit_depth: int
    xdec: int
    ydec: int
    nplanes: int
    frame_rate: Fraction


class Y4MReader:
    """Pull-model frame source over a .y4m file (g1s_y4m_*)."""

    def __init__(self, path: str):
        self._L = _lib.lib()
        err = C.create_string_buffer(512)
        self._h = self._L.g1s_y4m_open(str(path).encode(), err, len(err))
        if not self._h:
            raise ValueError(err.value.decode() or "cannot open y4m file")
        info = G1SY4MInfo()
        self._L.g1s_y4m_get_info(self._h, C.byref(info))
        self.details = VideoDetails(info.width, info.height, info.bit_depth, info.xdec, info.ydec, info.nplanes,
                                    Fraction(info.fps_num, info.fps_den))

    @property
    def handle(self) -> int:
        return self._h

    def get_frame(self) -> Optional[Sequence[np.ndarray]]:
        """The next frame as plane arrays (copies), or None at end of stream."""
        f = G1SFrame()
        rc = self._L.g1s_y4m_next(self._h, C.byref(f))
        if rc < 0:
            raise ValueError(self._L.g1s_y4m_last_error(self._h).decode())
        if rc == 0:
            return None
        d = self.details
        planes = []
        for c in range(d.nplanes):
            w = d.width if c == 0 else (d.width + (1 << d.xdec) - 1) >> d.xdec
            h = d.height if c == 0 else (d.height + (1 << d.ydec) - 1) >> d.ydec
            dt = np.uint16 if f.bytes_per_sample == 2 else np.uint8
            buf = C.cast(f.data[c], C.POINTER(C.c_uint8 * (h * f.stride_bytes[c]))).contents
            planes.append(np.frombuffer(buf, dtype=dt).reshape(h, -1)[:, :w].copy())
        return planes

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.g1s_y4m_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def diff_y4m_files(source: str, denoised: str, output: str, *, ar_coeff_lag: int = 3, luma_only: bool = False,
                   batch_frames: int = 0, device: int = -1, filters: Optional[str] = None,
                   devices: Optional[Sequence[int]] = None) -> Tuple[int, bool]:
    """`grav1synth diff SOURCE DENOISED -o OUTPUT [-f FILTERS]` for .y4m inputs.  Returns (frames, unequal).
    devices: HIP ordinals of a frame-sharded job (one generator each, the video dealt batch by batch; an ordinal may
    repeat); None = one generator on `device`."""
    L = _lib.lib()
    opts = G1SOpts(C.sizeof(G1SOpts), device, ar_coeff_lag, int(luma_only), batch_frames, 0)
    frames = C.c_uint64(0)
    unequal = C.c_int(0)
    err = C.create_string_buffer(512)
    if devices is not None:
        devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        rc = L.g1s_diff_y4m_files_sharded(str(source).encode(), str(denoised).encode(), str(output).encode(), C.byref(opts),
                                          filters.encode() if filters else None, devs, len(devices), C.byref(frames),
                                          C.byref(unequal), err, len(err))
    else:
        rc = L.g1s_diff_y4m_files_filtered(str(source).encode(), str(denoised).encode(), str(output).encode(), C.byref(opts),
                                           filters.encode() if filters else None, C.byref(frames), C.byref(unequal), err, len(err))
    if rc:
        raise RuntimeError(err.value.decode() or f"g1s_diff_y4m_files failed ({rc})")
    if unequal.value:
        log.warning(UNEQUAL_WARNING)
    log.info("Computed diff for %d frames", frames.value)
    return int(frames.value), bool(unequal.value)


def write_y4m(path: str, frames: Iterable[Sequence], bit_depth: int, xdec: int, ydec: int,
              fps: Fraction = Fraction(24, 1)) -> int:
    """Write planar frames (sequences of 2-D uint8 / uint16 arrays or tensors) as YUV4MPEG2."""
    n = 0
    with open(path, "wb") as f:
        for planes in frames:
            planes = [np.ascontiguousarray(p.cpu().numpy() if hasattr(p, "cpu") else p) for p in planes]
            if n == 0:
                h, w = planes[0].shape
                if len(planes) == 1:
                    cs = "mono" if bit_depth == 8 else f"mono{bit_depth}"
                else:
                    cs = {(1, 1): "420", (1, 0): "422", (0, 0): "444"}[(xdec, ydec)]
                    if bit_depth == 8 and cs == "420":
                        cs = "420jpeg"
                    if bit_depth > 8:
                        cs += f"p{bit_depth}"
                f.write(f"YUV4MPEG2 W{w} H{h} F{fps.numerator}:{fps.denominator} Ip A1:1 C{cs}\n".encode())
            f.write(b"FRAME\n")
            for p in planes:
                f.write(p.astype("<u2" if bit_depth > 8 else np.uint8, copy=False).tobytes())
            n += 1
    return n
