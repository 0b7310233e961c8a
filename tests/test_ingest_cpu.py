"""The caller side of the path without a GPU: the YUV4MPEG2 frame source (where the reference has its
libav reader, src/reader.rs:37-212) hands out exactly the planes that were written, for every format
the reference's reader accepts (8/10/12-bit, 4:2:0 / 4:2:2 / 4:4:4, src/reader.rs:51-85), signals
end of stream as `None`, and reports malformed files instead of guessing."""
from fractions import Fraction

import numpy as np
import pytest

from grav1synth_amd.ingest import Y4MReader, write_y4m


def _frames(n, w, h, bd, xdec, ydec, nplanes, seed=7):
    rng = np.random.default_rng(seed)
    dt = np.uint16 if bd > 8 else np.uint8
    out = []
    for _ in range(n):
        planes = [rng.integers(0, 1 << bd, (h, w), dtype=dt)]
        if nplanes == 3:
            cw, ch = (w + (1 << xdec) - 1) >> xdec, (h + (1 << ydec) - 1) >> ydec
            planes += [rng.integers(0, 1 << bd, (ch, cw), dtype=dt) for _ in range(2)]
        out.append(planes)
    return out


@pytest.mark.parametrize("bd,xdec,ydec,nplanes,w,h", [
    (8, 1, 1, 3, 64, 48), (10, 1, 1, 3, 50, 38), (12, 1, 0, 3, 48, 32), (10, 0, 0, 3, 33, 17),
    (8, 0, 0, 1, 40, 24), (8, 1, 1, 3, 35, 27),   # odd sizes: the file holds ceil(w/2) x ceil(h/2) chroma
    (10, 1, 1, 3, 2048, 1024),                     # 6 MB frames: the parallel positional-read path
])
def test_reader_returns_the_written_planes(tmp_path, bd, xdec, ydec, nplanes, w, h):
    frames = _frames(9 if w < 1000 else 6, w, h, bd, xdec, ydec, nplanes)   # more than the read-ahead ring holds
    path = tmp_path / "a.y4m"
    assert write_y4m(str(path), frames, bd, xdec, ydec, Fraction(30000, 1001)) == len(frames)
    r = Y4MReader(str(path))
    d = r.details
    assert (d.width, d.height, d.bit_depth, d.nplanes) == (w, h, bd, nplanes)
    if nplanes == 3:
        assert (d.xdec, d.ydec) == (xdec, ydec)
    assert d.frame_rate == Fraction(30000, 1001)
    for want in frames:
        got = r.get_frame()
        assert got is not None and len(got) == nplanes
        for a, b in zip(got, want):
            assert a.dtype == b.dtype and np.array_equal(a, b)
    assert r.get_frame() is None          # end of stream
    assert r.get_frame() is None          # and it stays there
    r.close()


def test_header_defaults_and_tags(tmp_path):
    p = tmp_path / "b.y4m"
    p.write_bytes(b"YUV4MPEG2 W16 H8 Ip A1:1 XYSCSS=420JPEG\nFRAME\n" + bytes(16 * 8 * 3 // 2))
    r = Y4MReader(str(p))
    d = r.details
    assert (d.width, d.height, d.bit_depth, d.xdec, d.ydec, d.nplanes) == (16, 8, 8, 1, 1, 3)
    assert d.frame_rate == Fraction(25, 1)     # the format's default when F is absent
    assert r.get_frame() is not None and r.get_frame() is None
    r.close()


def test_malformed_files_are_reported(tmp_path):
    p = tmp_path / "c.y4m"
    p.write_bytes(b"RIFF....")
    with pytest.raises(ValueError, match="not a YUV4MPEG2"):
        Y4MReader(str(p))
    p.write_bytes(b"YUV4MPEG2 W16 H8 F24:1 C411\n")
    with pytest.raises(ValueError, match="unsupported colour space"):
        Y4MReader(str(p))
    p.write_bytes(b"YUV4MPEG2 W0 H8 F24:1 C420\n")
    with pytest.raises(ValueError, match="bad header"):
        Y4MReader(str(p))
    with pytest.raises(ValueError, match="cannot open"):
        Y4MReader(str(tmp_path / "missing.y4m"))
    # a truncated frame is an error, not an end of stream
    p.write_bytes(b"YUV4MPEG2 W16 H8 F24:1 C420\nFRAME\n" + bytes(100))
    r = Y4MReader(str(p))
    with pytest.raises(ValueError, match="truncated frame 0"):
        r.get_frame()
    r.close()
    # the same on the big-frame (positional read) path: second frame cut short
    big = _frames(2, 2048, 1024, 10, 1, 1, 3)
    write_y4m(str(p), big, 10, 1, 1)
    data = p.read_bytes()
    p.write_bytes(data[:-1000])
    r = Y4MReader(str(p))
    assert r.get_frame() is not None
    with pytest.raises(ValueError, match="truncated frame 1"):
        r.get_frame()
    r.close()


def test_reader_takes_a_stream_from_a_pipe(tmp_path):
    """No libav in the image: what stands in for the reference's decoder is a decoder's output piped in --
    `ffmpeg -i in.mkv -f yuv4mpegpipe - > fifo`.  The reader must therefore work on something it cannot seek in or take the
    size of: a FIFO, written by another thread frame by frame, big frames included (the positional-read path needs a file)."""
    import os
    import threading

    for (w, h, bd) in ((64, 48, 8), (2048, 1024, 10)):
        frames = _frames(5, w, h, bd, 1, 1, 3, seed=11)
        plain = tmp_path / f"plain_{w}.y4m"
        write_y4m(str(plain), frames, bd, 1, 1, Fraction(24, 1))
        fifo = tmp_path / f"pipe_{w}.y4m"
        os.mkfifo(fifo)

        def feed():
            with open(plain, "rb") as src, open(fifo, "wb") as dst:  # (opening the FIFO blocks until the reader opens it)
                while True:
                    chunk = src.read(1 << 16)
                    if not chunk:
                        break
                    dst.write(chunk)

        t = threading.Thread(target=feed, daemon=True)
        t.start()
        r = Y4MReader(str(fifo))
        assert (r.details.width, r.details.height, r.details.bit_depth) == (w, h, bd)
        for want in frames:
            got = r.get_frame()
            assert got is not None
            for a, b in zip(got, want):
                assert np.array_equal(a, b)
        assert r.get_frame() is None
        r.close()
        t.join(timeout=10)
        assert not t.is_alive()
