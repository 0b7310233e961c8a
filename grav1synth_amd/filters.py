"""`--filters` of `grav1synth diff`: FilterChain (/root/reference/src/filters.rs) over libg1s_diff.so.

FilterChain(text) parses with the reference's grammar and error texts (src/filters.rs:16-110); `.filters` lists what
was parsed; `.apply(frame)` is FilterChain::apply (:112-116) on a frame descriptor: crop is extent arithmetic (views
into the same planes, host or device), resize is refused (FilterError, "not supported")."""
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

    def apply(self, frame):
        """frame: grav1synth_amd.diff.Frame; returns a Frame whose planes are views of the input's."""
        from .diff import Frame

        out = []
        planes = list(frame.planes)
        for f in self.filters:
            if isinstance(f, Resize):
                raise FilterError(f"resize:width={f.width},height={f.height},alg={f.alg} -- the resize filter is not supported "
                                  "here (crop is): resize the source before diff")
            h, w = planes[0].shape
            if f.left + f.right >= w or f.top + f.bottom >= h:
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
