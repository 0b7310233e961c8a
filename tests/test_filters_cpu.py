"""N3: the `--filters` parser and the front door's refusal rules, host side (no GPU).

The parser cases are the reference's own (FilterChain tests, /root/reference/src/filters.rs:199-363), one for one: same
inputs, same parsed values, same error fragments."""
import logging

import numpy as np
import pytest

from grav1synth_amd.diff import Frame
from grav1synth_amd.filters import Crop, FilterChain, FilterError, Resize


def _err(text):
    with pytest.raises(FilterError) as e:
        FilterChain(text)
    return str(e.value)


def test_new_accepts_empty_filter_chain():
    assert FilterChain("").filters == []


def test_new_parses_crop_filter_args():
    assert FilterChain("crop:top=1,bottom=2,left=3,right=4").filters == [Crop(1, 2, 3, 4)]


def test_new_parses_resize_filter_with_default_algorithm():
    assert FilterChain("resize:width=1920,height=1080").filters == [Resize(1920, 1080, "catmullrom")]


@pytest.mark.parametrize("alg", ["hermite", "catmullrom", "mitchell", "lanczos", "spline36"])
def test_new_parses_resize_filter_with_all_supported_algorithms(alg):
    assert FilterChain(f"resize:width=640,height=360,alg={alg}").filters == [Resize(640, 360, alg)]


def test_new_parses_multiple_filters_in_order():
    assert FilterChain("crop:top=4;resize:width=320,height=240,alg=lanczos").filters == [Crop(4, 0, 0, 0), Resize(320, 240, "lanczos")]


@pytest.mark.parametrize("text,fragment", [
    ("crop", 'Invalid filter syntax in "crop"'),
    ("rotate:degrees=90", 'Unrecognized filter "rotate"'),
    ("crop:top", 'Invalid filter syntax in "top"'),
    ("crop:width=12", 'Unrecognized crop arg "width"'),
    ("crop:top=abc", "invalid digit found in string"),
    ("resize:width=640,height", 'Invalid filter syntax in "height"'),
    ("resize:width=640,height=360,scale=2", 'Unrecognized resize arg "scale"'),
    ("resize:width=640,height=360,alg=nearest", 'Unrecognized resize algorithm "nearest"'),
    ("resize:width=640", "Both width and height must be provided to resize filter"),
    ("resize:height=360", "Both width and height must be provided to resize filter"),
    ("resize:width=wide,height=360", "invalid digit found in string"),
    ("resize:width=640,height=tall", "invalid digit found in string"),
    # beyond the reference's tests, same grammar: empty pieces, empty numbers, numbers past usize
    ("crop:top=1;", 'Invalid filter syntax in ""'),
    ("crop:", 'Invalid filter syntax in ""'),
    ("crop:top=", "cannot parse integer from empty string"),
    ("crop:top=99999999999999999999999", "number too large to fit in target type"),
    ("crop:top=-1", "invalid digit found in string"),
])
def test_new_rejects(text, fragment):
    assert fragment in _err(text)


def test_crop_is_extent_arithmetic_on_the_planes():
    y = np.arange(64 * 96, dtype=np.uint16).reshape(64, 96)
    u = np.arange(32 * 48, dtype=np.uint16).reshape(32, 48)
    f = FilterChain("crop:top=4,left=8;crop:bottom=2,right=6").apply(Frame([y, u, u + 1], 1, 1))
    assert f.planes[0].shape == (58, 82) and f.planes[1].shape == (29, 41)
    assert f.planes[0][0, 0] == y[4, 8] and f.planes[1][0, 0] == u[2, 4] and f.planes[2][-1, -1] == u[30, 44] + 1
    assert np.shares_memory(f.planes[0], y)  # views: no sample was copied


def test_crop_must_fall_on_chroma_samples_and_leave_something():
    y = np.zeros((64, 96), np.uint8)
    u = np.zeros((32, 48), np.uint8)
    with pytest.raises(FilterError, match="multiples of the chroma subsampling"):
        FilterChain("crop:left=3").apply(Frame([y, u, u], 1, 1))
    FilterChain("crop:left=3").apply(Frame([y], 1, 1))  # luma only: any amount
    with pytest.raises(FilterError, match="leaves nothing"):
        FilterChain("crop:top=32,bottom=32").apply(Frame([y, u, u], 1, 1))


def test_resize_taps_are_the_oracles_and_sum_to_one():
    """N3: the taps of an axis (g1s_resize_plan: host code of csrc/resize.hip, no device needed) are the ones
    oracle/resize_oracle.c forms -- the five kernels, up and down, mirrored edges -- and every output sample's taps sum to 1."""
    import ctypes as C

    from grav1synth_amd import _lib
    from tests.oracle_binding import resize_plan

    L = _lib.lib()
    for alg in ("hermite", "catmullrom", "mitchell", "lanczos", "spline36"):
        for src, dst in ((96, 48), (48, 96), (100, 37), (37, 100), (64, 64), (7, 50), (1920, 1280)):
            want_idx, want_coef = resize_plan(alg, src, dst)
            taps = C.c_uint32()
            idx = np.zeros(want_idx.shape, np.int32)
            coef = np.zeros(want_coef.shape, np.float32)
            assert L.g1s_resize_plan(alg.encode(), src, dst, C.byref(taps), idx.ctypes.data, coef.ctypes.data, idx.size) == 0
            assert taps.value == want_idx.shape[1] == 2 * int(np.ceil((2 if alg in ("hermite", "catmullrom", "mitchell") else 3) / min(dst / src, 1.0)))
            assert np.array_equal(idx, want_idx) and np.array_equal(coef.view(np.uint32), want_coef.view(np.uint32)), (alg, src, dst)
            assert np.allclose(coef.astype(np.float64).sum(axis=1), 1.0, atol=1e-6)
            assert idx.min() >= 0 and idx.max() <= src - 1
    assert L.g1s_resize_plan(b"bilinear", 8, 8, C.byref(C.c_uint32()), None, None, 0) != 0


def test_resize_oracle_properties():
    """oracle/resize_oracle.c on what must hold for any separable resampler with normalised taps: a constant plane stays
    constant, the same size with an interpolating kernel is the identity, values stay inside the bit depth, a horizontal
    ramp stays monotone when enlarged with a non-negative kernel."""
    from tests.oracle_binding import resize_planes

    rng = np.random.default_rng(5)
    const = np.full((40, 56), 700, np.uint16)
    for alg in ("hermite", "catmullrom", "mitchell", "lanczos", "spline36"):
        out = resize_planes([const], 0, 0, 84, 26, 10, alg)[0]
        assert out.shape == (26, 84) and (out == 700).all(), alg
    noise = rng.integers(0, 1024, (40, 56), dtype=np.uint16)
    for alg in ("hermite", "catmullrom", "lanczos", "spline36"):  # (interpolating kernels: 1 at 0, 0 at the other integers)
        assert np.array_equal(resize_planes([noise], 0, 0, 56, 40, 10, alg)[0], noise), alg
    hot = np.where(rng.random((40, 56)) < 0.5, 0, 1023).astype(np.uint16)  # (overshoot is clamped to the bit depth)
    out = resize_planes([hot], 0, 0, 112, 80, 10, "lanczos")[0]
    assert out.max() <= 1023
    ramp = np.tile(np.arange(64, dtype=np.uint8) * 4, (8, 1))
    out = resize_planes([ramp], 0, 0, 256, 8, 8, "hermite")[0].astype(int)
    assert (np.diff(out, axis=1) >= 0).all()
    y, u = np.zeros((64, 96), np.uint8), np.zeros((32, 48), np.uint8)
    o = resize_planes([y, u, u], 1, 1, 48, 32, 8)
    assert [p.shape for p in o] == [(32, 48), (16, 24), (16, 24)]


def test_front_door_refusals(tmp_path, caplog):
    """src/main.rs:354-394: each refusal is one logged line and a normal return; nothing is opened or written."""
    from grav1synth_amd import cli

    a, b, out = str(tmp_path / "a.y4m"), str(tmp_path / "b.y4m"), str(tmp_path / "o.tbl")
    with caplog.at_level(logging.INFO, logger="grav1synth"):
        assert cli.diff_command(a, b, a) == -1
        assert cli.diff_command(a, b, b) == -1
        assert caplog.records[-1].getMessage() == cli.SAME_AS_OUTPUT and caplog.records[-1].levelname == "ERROR"
        assert cli.diff_command(a, a, out) == -1
        assert caplog.records[-1].getMessage() == cli.SAME_INPUTS
        assert cli.diff_command(a, b, out, filters="crop:up=2") == -1
        assert caplog.records[-1].getMessage() == 'Invalid filter chain: Unrecognized crop arg "up"'
        open(out, "w").write("keep me")
        asked = []
        assert cli.diff_command(a, b, out, confirm=lambda p: asked.append(p) or False) == -1
        assert asked == [f"File {out} exists. Overwrite?"]
        assert caplog.records[-1].getMessage() == cli.NOT_OVERWRITING and caplog.records[-1].levelname == "WARNING"
        assert open(out).read() == "keep me"
    assert cli.main(["diff", a, a, "-o", out]) == 0  # a refusal is not a failure (`return Ok(())`)


def test_front_door_compares_paths_like_pathbuf_and_fails_without_a_terminal(tmp_path, monkeypatch):
    """the reference compares PathBuf values (src/main.rs:354, :362): `a//b`, `a/./b`, `a/b/` are the path `a/b`;
    and dialoguer's Confirm::interact()? is an error without a terminal -- a non-zero exit, not a quiet refusal."""
    import io

    from grav1synth_amd import cli

    d = str(tmp_path)
    assert cli._same_path(d + "/a.y4m", d + "//a.y4m") and cli._same_path(d + "/./a.y4m", d + "/a.y4m")
    assert cli._same_path("x/y/", "x/y") and not cli._same_path("./x", "x") and not cli._same_path("x/../x", "x")
    assert cli.diff_command(d + "/a.y4m", d + "/b.y4m", d + "//a.y4m") == -1
    assert cli.diff_command(d + "/a.y4m", d + "/./a.y4m", d + "/o.tbl") == -1
    out = str(tmp_path / "o.tbl")
    open(out, "w").write("keep me")
    monkeypatch.setattr("sys.stdin", io.StringIO(""))  # not a terminal
    assert cli.main(["diff", d + "/a.y4m", d + "/b.y4m", "-o", out]) == 1
    assert cli.main(["estimate", d + "/a.y4m", "-o", out]) == 1
    assert open(out).read() == "keep me"


def test_crop_amounts_that_would_wrap_a_sum_are_refused():
    """`crop:left=18446744073709551615,right=1` must not pass the bounds check by wrapping to 0."""
    y = np.zeros((64, 96), np.uint8)
    for chain in ("crop:left=18446744073709551615,right=1", "crop:top=18446744073709551615,bottom=1",
                  "crop:left=96", "crop:right=18446744073709551615"):
        with pytest.raises(FilterError, match="leaves nothing"):
            FilterChain(chain).apply(Frame([y], 1, 1))
    assert FilterChain("crop:left=95").apply(Frame([y], 1, 1)).planes[0].shape == (64, 1)
