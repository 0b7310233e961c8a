"""`--filters` of `grav1synth diff`: FilterChain (/root/reference/src/filters.rs) over libg1s_diff.so.

FilterChain(text) parses with the reference's grammar and error texts (src/filters.rs:16-110); `.filters` lists what
was parsed; `.apply(frame, bit_depth)` is FilterChain::apply (:112-116): crop is extent arithmetic (views into the same
planes, host or device); resize (:150-178) runs on the device (csrc/resize.hip: no CPU fallback) and returns host planes."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Union

from . import _lib
from ._lib import G1SFilterDesc


class FilterError(ValueError):
    """anyhow::Error of FilterChain::new / a filter this path does not serve."""


@dataclass(frozen=True)
class Crop:
    top: int = 0
    bottom: int = 0
    left: int = 0
    right: int = 0


@dataclass(frozen=True)
class Resize:
    width: int
    height: int
    alg: str = "catmullrom"


class FilterChain:
    def __init__(self, filters: str):
        L = _lib.lib()
        err = C.create_string_buffer(256)
        self._L = L
        self._h = L.g1s_filters_new(filters.encode(), err, len(err))
        if not self._h:
            raise FilterError(err.value.decode())
        self.filters: List[Union[Crop, Resize]] = []
        d = G1SFilterDesc()
        for i in range(L.g1s_filters_len(self._h)):
            L.g1s_filters_get(self._h, i, C.byref(d))
            self.filters.append(Crop(d.top, d.bottom, d.left, d.right) if d.kind == 0 else Resize(d.width, d.height, d.alg.decode()))

    @property
    def handle(self) -> int:
        return self._h

    def apply(self, frame, bit_depth: int = 0, device: int = -1):
        """frame: grav1synth_amd.diff.Frame.  Crops return views of the input's planes; a resize runs on the device and
        returns new host planes (numpy).  bit_depth: the source bit depth (needed by a resize of 16-bit samples)."""
        import numpy as np

        from .diff import Frame

        planes = list(frame.planes)
        for f in self.filters:
            if isinstance(f, Resize):
                hp = [np.ascontiguousarray(p.cpu().numpy() if hasattr(p, "cpu") else p) for p in planes]
                bd = bit_depth or (8 if hp[0].dtype == np.uint8 else 0)
                if not bd:
                    raise FilterError(f"resize:width={f.width},height={f.height},alg={f.alg} -- a resize needs the source bit depth")
                fr = Frame(hp, frame.xdec, frame.ydec).to_c([])
                outs = [np.zeros(((f.height >> frame.ydec) if c else f.height, (f.width >> frame.xdec) if c else f.width), hp[0].dtype)
                        for c in range(len(hp))]
                ptrs = (C.c_void_p * 3)(*[o.ctypes.data for o in outs] + [None] * (3 - len(outs)))
                strides = (C.c_size_t * 3)(*[o.strides[0] for o in outs] + [0] * (3 - len(outs)))
                err = C.create_string_buffer(256)
                rc = self._L.g1s_resize_frame_to_host(f.alg.encode(), C.byref(fr), bd, f.width, f.height, device, ptrs, strides, err, len(err))
                if rc:
                    raise FilterError(err.value.decode() or f"resize failed ({rc})")
                planes = outs
                continue
            h, w = planes[0].shape
            if f.left >= w or f.right >= w - f.left or f.top >= h or f.bottom >= h - f.top:
                raise FilterError(f"crop leaves nothing of a {w}x{h} frame")
            mx, my = ((1 << frame.xdec) - 1, (1 << frame.ydec) - 1) if len(planes) == 3 else (0, 0)
            if (f.left & mx) or (f.right & mx) or (f.top & my) or (f.bottom & my):
                raise FilterError("crop amounts must be multiples of the chroma subsampling")
            out = []
            for c, p in enumerate(planes):
                sx, sy = (frame.xdec, frame.ydec) if c else (0, 0)
                ph, pw = p.shape
                out.append(p[f.top >> sy: ph - (f.bottom >> sy), f.left >> sx: pw - (f.right >> sx)])
            planes = out
        return Frame(planes, frame.xdec, frame.ydec)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.g1s_filters_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
