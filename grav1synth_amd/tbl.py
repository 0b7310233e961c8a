"""`.tbl` film-grain table reader (the consumer side of `diff`'s output).

Format as written by the reference at src/main.rs:525-529,631-696 and read back
by `apply` through `av1_grain::parse_grain_table` (src/main.rs:228-241):

    filmgrn1
    E <start> <end> <apply_grain=1> <random_seed> <update_parameters=1>
    \tp <ar_coeff_lag> <ar_coeff_shift> <grain_scale_shift> <scaling_shift>
        <chroma_scaling_from_luma> <overlap_flag> <cb_mult> <cb_luma_mult>
        <cb_offset> <cr_mult> <cr_luma_mult> <cr_offset>
    \tsY <n>  <x y>*n      \tsCb <n> <x y>*n      \tsCr <n> <x y>*n
    \tcY <c>*(2*lag*(lag+1))   \tcCb <c>*(that+1)   \tcCr <c>*(that+1)

Host logic only (no GPU).  The writer is the C function g1s_format_tbl.
"""
from __future__ import annotations

from typing import List

from .diff import GrainTableSegment


class TblError(ValueError):
    pass


def parse_tbl(data: bytes) -> List[GrainTableSegment]:
    text = data.decode("ascii")
    lines = text.split("\n")
    if not lines or lines[0].strip() != "filmgrn1":
        raise TblError("missing filmgrn1 header")
    segs: List[GrainTableSegment] = []
    i = 1
    while i < len(lines):
        ln = lines[i]
        if not ln.strip():
            i += 1
            continue
        if not ln.startswith("E"):
            raise TblError(f"line {i + 1}: expected an E line")
        e = ln.split()
        if len(e) != 6:
            raise TblError(f"line {i + 1}: E line needs 5 fields")
        start, end, apply_grain, seed, update = (int(v) for v in e[1:])
        if apply_grain != 1 or update != 1:
            raise TblError(f"line {i + 1}: apply_grain/update_parameters must be 1")
        body = {}
        for k in range(1, 8):
            if i + k >= len(lines):
                raise TblError("truncated segment")
            t = lines[i + k].split()
            if not t:
                raise TblError("truncated segment")
            body[t[0]] = [int(v) for v in t[1:]]
        i += 8
        for key in ("p", "sY", "sCb", "sCr", "cY", "cCb", "cCr"):
            if key not in body:
                raise TblError(f"segment starting at {start}: missing {key} line")
        p = body["p"]
        if len(p) != 12:
            raise TblError("p line needs 12 fields")
        lag = p[0]
        if not 0 <= lag <= 3:
            raise TblError("ar_coeff_lag out of range")

        def points(key, cap):
            v = body[key]
            n = v[0]
            if n > cap or len(v) != 1 + 2 * n:
                raise TblError(f"{key}: bad point count")
            return [(v[1 + 2 * j], v[2 + 2 * j]) for j in range(n)]

        ncoef = 2 * lag * (lag + 1)
        if len(body["cY"]) != ncoef or len(body["cCb"]) != ncoef + 1 or len(body["cCr"]) != ncoef + 1:
            raise TblError("coefficient count does not match ar_coeff_lag")
        segs.append(GrainTableSegment(
            random_seed=seed, start_time=start, end_time=end,
            scaling_points_y=points("sY", 14), scaling_points_cb=points("sCb", 10),
            scaling_points_cr=points("sCr", 10),
            scaling_shift=p[3], ar_coeff_lag=lag,
            ar_coeffs_y=body["cY"], ar_coeffs_cb=body["cCb"], ar_coeffs_cr=body["cCr"],
            ar_coeff_shift=p[1], cb_mult=p[6], cb_luma_mult=p[7], cb_offset=p[8],
            cr_mult=p[9], cr_luma_mult=p[10], cr_offset=p[11],
            chroma_scaling_from_luma=bool(p[4]), grain_scale_shift=p[2], overlap_flag=bool(p[5])))
    return segs


def parse_tbl_native(data: bytes) -> List[GrainTableSegment]:
    """The same through the library's C parser (g1s_parse_tbl)."""
    import ctypes as C

    from . import _lib
    from ._lib import G1SSegment

    L = _lib.lib()
    n = C.c_size_t(0)
    err = C.create_string_buffer(256)
    cap = 64
    while True:
        buf = (G1SSegment * cap)()
        rc = L.g1s_parse_tbl(data, len(data), buf, cap, C.byref(n), err, len(err))
        if rc == -8:  # G1S_ERR_CAPACITY: *n_out holds the count
            cap = n.value
            continue
        if rc:
            raise TblError(err.value.decode() or f"g1s_parse_tbl failed ({rc})")
        return [GrainTableSegment.from_c(buf[i]) for i in range(n.value)]


class GrainTable:
    """Segments + the lookup `apply` does per frame (src/parser/frame.rs:617-633): the first segment with
    start_time <= ts < end_time; every hit advances that segment's seed by DEFAULT_GRAIN_SEED (wrapping)."""

    def __init__(self, segments: List[GrainTableSegment]):
        import ctypes as C

        from ._lib import G1SSegment

        self._n = len(segments)
        self._buf = (G1SSegment * max(self._n, 1))()
        for i, s in enumerate(segments):
            self._buf[i] = s.to_c()

    def segment_for(self, packet_ts: int):
        from . import _lib

        i = _lib.lib().g1s_tbl_segment_for(self._buf, self._n, int(packet_ts))
        return None if i < 0 else GrainTableSegment.from_c(self._buf[i])
