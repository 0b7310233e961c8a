"""Host-side mirror of `av1_grain::DiffGenerator` over the C-ABI.

Same three methods, argument meaning and error behaviour as the object the
reference drives at src/main.rs:420-427 (`new`), :442 (`diff_frame`, errors
propagate) and :524 (`finish`, consumes the generator), plus the `.tbl` writer
of src/main.rs:525-529,631-696.  All pixel work happens in the HIP kernels of
libg1s_diff.so; this module only marshals pointers.
"""
from __future__ import annotations

import ctypes as C
from collections import deque
from dataclasses import dataclass, field
from fractions import Fraction
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib
from ._lib import G1SError, G1SFrame, G1SOpts, G1SSegment, G1SStats

try:  # torch is plumbing (device memory); numpy host frames work without it
    import torch
except Exception:  # pragma: no cover
    torch = None

DEFAULT_GRAIN_SEED = 10956  # av1_grain::DEFAULT_GRAIN_SEED (src/parser/frame.rs:3)


@dataclass
class GrainTableSegment:
    """Field-for-field mirror of av1_grain::GrainTableSegment as read by
    src/parser/grain.rs:108-133."""

    random_seed: int
    start_time: int
    end_time: int
    scaling_points_y: List[Tuple[int, int]]
    scaling_points_cb: List[Tuple[int, int]]
    scaling_points_cr: List[Tuple[int, int]]
    scaling_shift: int
    ar_coeff_lag: int
    ar_coeffs_y: List[int]
    ar_coeffs_cb: List[int]
    ar_coeffs_cr: List[int]
    ar_coeff_shift: int
    cb_mult: int
    cb_luma_mult: int
    cb_offset: int
    cr_mult: int
    cr_luma_mult: int
    cr_offset: int
    chroma_scaling_from_luma: bool
    grain_scale_shift: int
    overlap_flag: bool

    @staticmethod
    def from_c(s: G1SSegment) -> "GrainTableSegment":
        return GrainTableSegment(
            random_seed=s.random_seed,
            start_time=s.start_time,
            end_time=s.end_time,
            scaling_points_y=[(s.scaling_points_y[i][0], s.scaling_points_y[i][1]) for i in range(s.num_y_points)],
            scaling_points_cb=[(s.scaling_points_cb[i][0], s.scaling_points_cb[i][1]) for i in range(s.num_cb_points)],
            scaling_points_cr=[(s.scaling_points_cr[i][0], s.scaling_points_cr[i][1]) for i in range(s.num_cr_points)],
            scaling_shift=s.scaling_shift,
            ar_coeff_lag=s.ar_coeff_lag,
            ar_coeffs_y=[s.ar_coeffs_y[i] for i in range(s.num_y_coeffs)],
            ar_coeffs_cb=[s.ar_coeffs_cb[i] for i in range(s.num_uv_coeffs)],
            ar_coeffs_cr=[s.ar_coeffs_cr[i] for i in range(s.num_uv_coeffs)],
            ar_coeff_shift=s.ar_coeff_shift,
            cb_mult=s.cb_mult,
            cb_luma_mult=s.cb_luma_mult,
            cb_offset=s.cb_offset,
            cr_mult=s.cr_mult,
            cr_luma_mult=s.cr_luma_mult,
            cr_offset=s.cr_offset,
            chroma_scaling_from_luma=bool(s.chroma_scaling_from_luma),
            grain_scale_shift=s.grain_scale_shift,
            overlap_flag=bool(s.overlap_flag),
        )

    def to_c(self) -> G1SSegment:
        s = G1SSegment()
        s.start_time, s.end_time, s.random_seed = self.start_time, self.end_time, self.random_seed
        s.num_y_points, s.num_cb_points, s.num_cr_points = (
            len(self.scaling_points_y), len(self.scaling_points_cb), len(self.scaling_points_cr))
        for dst, src in ((s.scaling_points_y, self.scaling_points_y), (s.scaling_points_cb, self.scaling_points_cb),
                         (s.scaling_points_cr, self.scaling_points_cr)):
            for i, (x, y) in enumerate(src):
                dst[i][0], dst[i][1] = x, y
        s.scaling_shift, s.ar_coeff_lag = self.scaling_shift, self.ar_coeff_lag
        s.num_y_coeffs, s.num_uv_coeffs = len(self.ar_coeffs_y), len(self.ar_coeffs_cb)
        for i, v in enumerate(self.ar_coeffs_y):
            s.ar_coeffs_y[i] = v
        for i, v in enumerate(self.ar_coeffs_cb):
            s.ar_coeffs_cb[i] = v
        for i, v in enumerate(self.ar_coeffs_cr):
            s.ar_coeffs_cr[i] = v
        s.ar_coeff_shift = self.ar_coeff_shift
        s.cb_mult, s.cb_luma_mult, s.cb_offset = self.cb_mult, self.cb_luma_mult, self.cb_offset
        s.cr_mult, s.cr_luma_mult, s.cr_offset = self.cr_mult, self.cr_luma_mult, self.cr_offset
        s.chroma_scaling_from_luma = int(self.chroma_scaling_from_luma)
        s.grain_scale_shift = self.grain_scale_shift
        s.overlap_flag = int(self.overlap_flag)
        return s


Plane = Union[np.ndarray, "torch.Tensor"]


@dataclass
class Frame:
    """A decoded frame: 1 or 3 planes (Y, U, V), u8 or u16 samples, like the
    `v_frame::Frame<T>` built at src/reader.rs:183-209.  Planes may be numpy
    arrays (host) or torch tensors (host or HIP device).

    Host planes are copied before diff_frame returns (the `&Frame` borrow of the reference).  `async_host=True`
    (pinned torch tensors only) queues the copies on the upload stream instead and returns at once: the caller
    must not touch the planes until `DiffGenerator.frames_copied()` covers the frame."""

    planes: Sequence[Plane]
    xdec: int = 1
    ydec: int = 1
    async_host: bool = False

    def to_c(self, keep: list) -> G1SFrame:
        f = G1SFrame()
        p0 = self.planes[0]
        f.height, f.width = int(p0.shape[0]), int(p0.shape[1])
        f.xdec, f.ydec = self.xdec, self.ydec
        f.nplanes = len(self.planes)
        on_dev = None
        for i, p in enumerate(self.planes):
            if torch is not None and isinstance(p, torch.Tensor):
                if p.stride(1) != 1:
                    raise ValueError("plane rows must be contiguous")
                isz = p.element_size()
                f.data[i] = p.data_ptr()
                f.stride_bytes[i] = p.stride(0) * isz
                if self.async_host and not p.is_cuda and not p.is_pinned():
                    raise ValueError("async_host needs pinned host tensors")
                # (a pinned tensor is usually a staging buffer its owner refills: synchronous unless asked otherwise)
                dev = 1 if p.is_cuda else (2 if self.async_host else 0)
            else:
                p = np.asarray(p)
                if p.strides[1] != p.dtype.itemsize:
                    raise ValueError("plane rows must be contiguous")
                isz = p.dtype.itemsize
                f.data[i] = p.ctypes.data
                f.stride_bytes[i] = p.strides[0]
                if self.async_host:
                    raise ValueError("async_host needs pinned host tensors")
                dev = 0
            if on_dev is None:
                on_dev = dev
            elif on_dev != dev:
                raise ValueError("all planes of a frame must live on the same side (host or device)")
            f.bytes_per_sample = isz
            keep.append(p)
        f.on_device = int(on_dev or 0)
        return f


def _as_frame(x, xdec, ydec) -> Frame:
    return x if isinstance(x, Frame) else Frame(list(x), xdec, ydec)


class DiffGenerator:
    """`av1_grain::DiffGenerator` on an MI355X.

    >>> differ = DiffGenerator(Fraction(24000, 1001), 10, 10)
    >>> differ.diff_frame(source_frame, denoised_frame)   # per frame pair, in order
    >>> segments = differ.finish()                        # consumes the generator
    """

    def __init__(self, fps, source_bit_depth: int, denoised_bit_depth: int, *, ar_coeff_lag: int = 3,
                 luma_only: bool = False, device: int = -1, batch_frames: int = 0, records_only=False):
        """records_only: False / 0 = fold locally; True / 1 = keep per-frame records (frame-shard mode);
        2 = keep per-frame latest states (frame-shard mode, per-frame half of the fold done here)."""
        self._L = _lib.lib()
        fr = Fraction(fps)
        opts = G1SOpts()
        opts.struct_size = C.sizeof(G1SOpts)
        opts.device = device
        opts.ar_coeff_lag = ar_coeff_lag
        opts.luma_only = int(luma_only)
        opts.batch_frames = batch_frames
        opts.records_only = int(records_only)
        self._h = self._L.g1s_diff_new(fr.numerator, fr.denominator, source_bit_depth, denoised_bit_depth,
                                       C.byref(opts))
        if not self._h:
            raise G1SError(-5, self._L.g1s_last_global_error().decode())
        # device-resident inputs must outlive the kernels that read them: (frames handed over so far, objects), pruned
        # by g1s_diff_frames_released -- a long stream pins a few batches, not every frame until finish
        self._keep: deque = deque()
        self._fed = 0
        self._finished = False
        self.records_only = records_only
        self.fps = fr
        self.ar_coeff_lag = ar_coeff_lag

    # -- DiffGenerator::diff_frame (src/main.rs:442) ------------------------------------
    def diff_frame(self, source, denoised, xdec: int = 1, ydec: int = 1, sync_torch: bool = True) -> None:
        s = _as_frame(source, xdec, ydec)
        d = _as_frame(denoised, xdec, ydec)
        keep: list = []
        fs, fd = s.to_c(keep), d.to_c(keep)
        if (fs.on_device or fd.on_device):
            if sync_torch and (fs.on_device == 1 or fd.on_device == 1):
                torch.cuda.current_stream().synchronize()  # producer (torch) -> consumer (engine stream)
            self._keep.append((self._fed + 1, keep))  # device planes must outlive the queued kernels
        self._fed += 1
        self._check(self._L.g1s_diff_frame(self._h, C.byref(fs), C.byref(fd)))
        self._prune()

    def _prune(self) -> None:
        if self._keep:
            done = int(self._L.g1s_diff_frames_released(self._h))
            while self._keep and self._keep[0][0] <= done:
                self._keep.popleft()

    def frames_copied(self, wait_for: int = 0) -> int:
        """g1s_diff_frames_copied: frame pairs whose host planes (Frame(..., async_host=True)) have been copied -- their
        buffers may be refilled.  wait_for = k blocks until frame pair k (1-based count) has been copied."""
        return int(self._L.g1s_diff_frames_copied(self._h, int(wait_for)))

    def frames_released(self) -> int:
        """g1s_diff_frames_released: frame pairs whose planes the generator no longer reads."""
        return int(self._L.g1s_diff_frames_released(self._h))

    # -- many frame pairs in one FFI call (g1s_diff_frames) -------------------------------
    @staticmethod
    def prepare_frames(pairs, xdec: int = 1, ydec: int = 1) -> "PreparedFrames":
        """Marshal (source, denoised) pairs once; the result can be fed to
        `diff_prepared` of any generator of the same format, any number of times."""
        keep: list = []
        n = len(pairs)
        src = (G1SFrame * n)()
        den = (G1SFrame * n)()
        for i, (s, d) in enumerate(pairs):
            src[i] = _as_frame(s, xdec, ydec).to_c(keep)
            den[i] = _as_frame(d, xdec, ydec).to_c(keep)
        return PreparedFrames(src, den, n, keep)

    def diff_prepared(self, prepared: "PreparedFrames", sync_torch: bool = True) -> None:
        if sync_torch and torch is not None and torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()
        self._keep.append((self._fed + prepared.n, prepared))
        self._fed += prepared.n
        self._check(self._L.g1s_diff_frames(self._h, prepared.src, prepared.den, prepared.n))
        self._prune()

    def sync(self) -> None:
        self._check(self._L.g1s_diff_sync(self._h))
        self._keep.clear()

    # -- DiffGenerator::finish (src/main.rs:524) ----------------------------------------
    def finish(self) -> List[GrainTableSegment]:
        cap = 64
        arr = (G1SSegment * cap)()
        n = C.c_size_t()
        rc = self._L.g1s_diff_finish(self._h, arr, cap, C.byref(n))
        if rc == _lib.G1S_ERR_CAPACITY:  # the segments stay in the generator: ask again with room for all of them
            cap = n.value
            arr = (G1SSegment * cap)()
            rc = self._L.g1s_diff_finish(self._h, arr, cap, C.byref(n))
        self._check(rc)
        self._finished = True
        self._keep.clear()
        return [GrainTableSegment.from_c(arr[i]) for i in range(n.value)]

    # -- frame-shard mode -----------------------------------------------------------------
    def take_records(self, width: int, height: int, nplanes: int, max_frames: int) -> Tuple[np.ndarray, int]:
        rs = self._L.g1s_record_size(width, height, 0, 0, nplanes, self.ar_coeff_lag)
        buf = np.zeros(rs * max(max_frames, 1), dtype=np.uint8)
        n = C.c_size_t()
        self._check(self._L.g1s_diff_take_records(self._h, buf.ctypes.data, buf.size, C.byref(n)))
        self._keep.clear()
        return buf[: rs * n.value].reshape(n.value, rs), n.value

    def take_latest(self, max_frames: int, sync: bool = False) -> np.ndarray:
        """records_only == 2: the latest states [n, g1s_latest_size] of the frames whose batches are
        complete (sync=False) or of everything queued so far (sync=True)."""
        bs = self._L.g1s_latest_size(self.ar_coeff_lag)
        buf = np.zeros(bs * max(max_frames, 1), dtype=np.uint8)
        n = C.c_size_t()
        self._check(self._L.g1s_diff_take_latest(self._h, int(sync), buf.ctypes.data, buf.size, C.byref(n)))
        if sync:
            self._keep.clear()
        return buf[: bs * n.value].reshape(n.value, bs)

    # -- measurement / parity hooks ------------------------------------------------------
    def set_timing(self, enable) -> None:
        """False / True: an event in front of every kernel of a batch (kernel_times); 2: one pair of events around the batch's
        whole chain of kernels (stats().ms_chain / chain_batches)."""
        self._L.g1s_diff_set_timing(self._h, int(enable))

    def kernel_times(self) -> dict:
        """Timed batches (set_timing): kernel name -> (milliseconds, launches), HIP events around each launch."""
        buf = C.create_string_buffer(1 << 14)
        n = self._L.g1s_diff_kernel_times(self._h, buf, len(buf))
        if n < 0:
            self._check(int(n))
        out = {}
        for line in buf.raw[:n].decode().splitlines():
            name, ms, launches = line.split("\t")
            out[name] = (float(ms), int(launches))
        return out

    def set_flat_finder(self, mode) -> None:
        """0 / False (default): certified fast path; 1 / True: the literal f64 evaluation of every block,
        one lane per block; 2: of every block, one wave per block."""
        self._check(self._L.g1s_diff_set_flat_finder(self._h, int(mode)))

    def stats(self) -> G1SStats:
        st = G1SStats()
        self._L.g1s_diff_get_stats(self._h, C.byref(st))
        return st

    def last_record(self) -> "Record":
        buf = np.zeros(64 << 20, dtype=np.uint8)
        self._check(self._L.g1s_diff_last_record(self._h, buf.ctypes.data, buf.size))
        return Record(buf)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.g1s_diff_free(self._h)
            self._h = None
            self._keep.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise G1SError(rc, self._L.g1s_diff_last_error(self._h).decode())


@dataclass
class PreparedFrames:
    src: object
    den: object
    n: int
    keep: list


class Record:
    """Read-only view of one per-frame integer record."""

    def __init__(self, buf: np.ndarray):
        self._L = _lib.lib()
        self.buf = buf
        nbw, nbh, npl, lag = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        rc = self._L.g1s_record_geometry(buf.ctypes.data, C.byref(nbw), C.byref(nbh), C.byref(npl), C.byref(lag))
        if rc != 0:
            raise G1SError(rc, "bad record")
        self.nbw, self.nbh, self.nplanes, self.lag = nbw.value, nbh.value, npl.value, lag.value

    @staticmethod
    def blank(width: int, height: int, xdec: int, ydec: int, nplanes: int, lag: int) -> "Record":
        L = _lib.lib()
        size = L.g1s_record_size(width, height, xdec, ydec, nplanes, lag)
        buf = np.zeros(size, dtype=np.uint8)
        rc = L.g1s_record_init(buf.ctypes.data, buf.size, width, height, xdec, ydec, nplanes, lag)
        if rc != 0:
            raise G1SError(rc, "g1s_record_init")
        return Record(buf)

    def views(self, c: int):
        """Writable numpy views (mask, scores, S, Sb+nobs, luma_sum, sum_d, sum_d2)
        into the record buffer -- used to assemble records in tests."""
        d = self.buf.ctypes.data
        nb = self.nbw * self.nbh
        S, Sb, nobs = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)(), C.c_int64()
        n = self._L.g1s_record_ar_sums(d, c, C.byref(S), C.byref(Sb), C.byref(nobs))
        ls, sd, sd2 = C.POINTER(C.c_uint32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_uint32)()
        self._L.g1s_record_block_stats(d, c, C.byref(ls), C.byref(sd), C.byref(sd2))
        return dict(
            mask=np.ctypeslib.as_array(self._L.g1s_record_flat_mask(d), shape=(nb,)),
            scores=np.ctypeslib.as_array(self._L.g1s_record_scores(d), shape=(nb,)),
            S=np.ctypeslib.as_array(S, shape=(n, n)),
            Sb_nobs=np.ctypeslib.as_array(Sb, shape=(n + 1,)),
            luma_sum=np.ctypeslib.as_array(ls, shape=(nb,)),
            sum_d=np.ctypeslib.as_array(sd, shape=(nb,)),
            sum_d2=np.ctypeslib.as_array(sd2, shape=(nb,)),
        )

    def flat_mask(self) -> np.ndarray:
        p = self._L.g1s_record_flat_mask(self.buf.ctypes.data)
        return np.ctypeslib.as_array(p, shape=(self.nbh, self.nbw)).copy()

    def scores(self) -> np.ndarray:
        p = self._L.g1s_record_scores(self.buf.ctypes.data)
        return np.ctypeslib.as_array(p, shape=(self.nbh, self.nbw)).copy()

    def ar_sums(self, c: int):
        S, Sb, nobs = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)(), C.c_int64()
        n = self._L.g1s_record_ar_sums(self.buf.ctypes.data, c, C.byref(S), C.byref(Sb), C.byref(nobs))
        if n < 0:
            raise G1SError(n, "bad plane")
        return (np.ctypeslib.as_array(S, shape=(n, n)).copy(), np.ctypeslib.as_array(Sb, shape=(n,)).copy(),
                nobs.value)

    def block_stats(self, c: int):
        ls, sd, sd2 = C.POINTER(C.c_uint32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_uint32)()
        nb = self._L.g1s_record_block_stats(self.buf.ctypes.data, c, C.byref(ls), C.byref(sd), C.byref(sd2))
        if nb < 0:
            raise G1SError(nb, "bad plane")
        return (np.ctypeslib.as_array(ls, shape=(nb,)).copy(), np.ctypeslib.as_array(sd, shape=(nb,)).copy(),
                np.ctypeslib.as_array(sd2, shape=(nb,)).copy())


class RecordFold:
    """The ordered fold over records (host only): what runs after the RCCL
    exchange in frame-shard mode, and inside DiffGenerator on a single GPU."""

    def __init__(self, fps, ar_coeff_lag: int = 3):
        self._L = _lib.lib()
        fr = Fraction(fps)
        self._h = self._L.g1s_fold_new(fr.numerator, fr.denominator, ar_coeff_lag)
        if not self._h:
            raise G1SError(-1, "g1s_fold_new failed")

    def push(self, record: np.ndarray) -> None:
        record = np.ascontiguousarray(record, dtype=np.uint8)
        rc = self._L.g1s_fold_push(self._h, record.ctypes.data, record.size)
        if rc != 0:
            raise G1SError(rc, self._L.g1s_fold_last_error(self._h).decode())

    def push_many(self, records: np.ndarray) -> None:
        """records: [n, record_size] uint8, frame order."""
        records = np.ascontiguousarray(records, dtype=np.uint8)
        if records.shape[0] == 0:
            return
        rc = self._L.g1s_fold_push_many(self._h, records.ctypes.data, records.shape[1], records.shape[0])
        if rc != 0:
            raise G1SError(rc, self._L.g1s_fold_last_error(self._h).decode())

    def push_latest_many(self, blobs: np.ndarray) -> None:
        """blobs: [n, g1s_latest_size] uint8 latest states, frame order: the ordered half only."""
        blobs = np.ascontiguousarray(blobs, dtype=np.uint8)
        if blobs.shape[0] == 0:
            return
        rc = self._L.g1s_fold_push_latest(self._h, blobs.ctypes.data, blobs.shape[1], blobs.shape[0])
        if rc != 0:
            raise G1SError(rc, self._L.g1s_fold_last_error(self._h).decode())

    def finish(self) -> List[GrainTableSegment]:
        cap = 64
        arr = (G1SSegment * cap)()
        n = C.c_size_t()
        rc = self._L.g1s_fold_finish(self._h, arr, cap, C.byref(n))
        if rc == _lib.G1S_ERR_CAPACITY:
            cap = n.value
            arr = (G1SSegment * cap)()
            rc = self._L.g1s_fold_finish(self._h, arr, cap, C.byref(n))
        if rc != 0:
            raise G1SError(rc, self._L.g1s_fold_last_error(self._h).decode())
        return [GrainTableSegment.from_c(arr[i]) for i in range(n.value)]

    def close(self):
        if getattr(self, "_h", None):
            self._L.g1s_fold_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def latest_size(ar_coeff_lag: int = 3) -> int:
    return int(_lib.lib().g1s_latest_size(ar_coeff_lag))


def latest_from_records(records: np.ndarray, ar_coeff_lag: int = 3) -> np.ndarray:
    """The per-frame half of the fold on the host: [n, record_size] records -> [n, latest_size] blobs."""
    L = _lib.lib()
    records = np.ascontiguousarray(records, dtype=np.uint8)
    bs = int(L.g1s_latest_size(ar_coeff_lag))
    out = np.zeros((records.shape[0], bs), dtype=np.uint8)
    rc = L.g1s_latest_from_records(records.ctypes.data, records.shape[1], records.shape[0], ar_coeff_lag, out.ctypes.data, bs)
    if rc != 0:
        raise G1SError(rc, "g1s_latest_from_records failed")
    return out


def format_tbl(segments: Sequence[GrainTableSegment]) -> bytes:
    """`filmgrn1` text exactly as src/main.rs:525-529,631-696 writes it."""
    L = _lib.lib()
    arr = (G1SSegment * max(len(segments), 1))(*[s.to_c() for s in segments])
    buf = C.create_string_buffer(1024 + 2048 * len(segments))
    n = L.g1s_format_tbl(arr, len(segments), buf, len(buf))
    if n < 0:
        raise G1SError(n, "g1s_format_tbl")
    return buf.raw[:n]


def write_tbl(path: str, segments: Sequence[GrainTableSegment]) -> None:
    with open(path, "wb") as f:
        f.write(format_tbl(segments))
