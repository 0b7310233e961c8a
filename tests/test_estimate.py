"""N4: the single-source noise estimator of `grav1synth estimate` (/root/reference/src/main.rs:534-608).

CPU: the oracle's scalar restatement (oracle/estimate_oracle.c) against an independent vectorised numpy derivation, the
command's output format, the front door's refusals.  GPU (`-m gpu`): the HIP kernel's per-frame estimates against the
oracle, bit for bit (the integer sums are exact; the f64 is formed with the same three operations), on odd sizes, every
depth, strided device planes, host planes, noise-free and all-edge content (None)."""
import logging
import math

import numpy as np
import pytest

from grav1synth_amd.synth import SynthSpec, make_pair
from tests.oracle_binding import estimate_plane_noise


def _numpy_estimate(p: np.ndarray, bd: int):
    """sum over the interior of [Sobel magnitude < 50] * |Laplacian|, both rounded to 8-bit scale: array form."""
    a = p.astype(np.int64)
    sh = bd - 8
    half = (1 << sh) >> 1
    m = lambda dy, dx: a[1 + dy: a.shape[0] - 1 + dy, 1 + dx: a.shape[1] - 1 + dx]  # noqa: E731
    gx = (m(-1, -1) - m(-1, 1)) + (m(1, -1) - m(1, 1)) + 2 * (m(0, -1) - m(0, 1))
    gy = (m(-1, -1) - m(1, -1)) + (m(-1, 1) - m(1, 1)) + 2 * (m(-1, 0) - m(1, 0))
    ga = (np.abs(gx) + np.abs(gy) + half) >> sh
    v = 4 * m(0, 0) - 2 * (m(-1, 0) + m(1, 0) + m(0, -1) + m(0, 1)) + (m(-1, -1) + m(-1, 1) + m(1, -1) + m(1, 1))
    on = ga < 50
    accum = int((((np.abs(v) + half) >> sh) * on).sum())
    count = int(on.sum())
    return None if count < 16 else accum / (6 * count) * 1.2533141373155003


def _plane(spec, k):
    s, _ = make_pair(spec, k)
    return s[0].numpy()


@pytest.mark.parametrize("spec", [SynthSpec(320, 192, 8), SynthSpec(326, 198, 10), SynthSpec(97, 61, 12), SynthSpec(3, 3, 8, textured=False)],
                         ids=lambda s: f"{s.width}x{s.height}_{s.bit_depth}b")
def test_oracle_estimator_equals_the_numpy_derivation(spec):
    for k in range(2):
        p = _plane(spec, k)
        assert estimate_plane_noise(p, spec.bit_depth) == _numpy_estimate(p, spec.bit_depth)


def test_oracle_estimator_edge_cases():
    flat = np.full((64, 64), 100, np.uint8)
    assert estimate_plane_noise(flat, 8) == 0.0                      # smooth everywhere, no noise
    assert estimate_plane_noise(np.zeros((2, 50), np.uint8), 8) is None   # no interior pixel
    checker = ((np.indices((40, 40)).sum(0) & 1) * 255).astype(np.uint8)
    noisy = checker.copy()
    assert estimate_plane_noise(noisy[:, :1].repeat(40, 1), 8) is not None     # vertical stripes of one value per row: smooth rows
    edges = (np.indices((40, 40))[1] // 2 % 2 * 255).astype(np.uint8)          # hard vertical edges everywhere
    assert estimate_plane_noise(edges, 8) is None
    # a known value: +-1 salt on a flat field
    p = np.full((32, 32), 50, np.uint16)
    p[10, 10] = 54
    got = estimate_plane_noise(p, 10)
    assert got == _numpy_estimate(p, 10) and got > 0


def test_format_matches_the_commands_output():
    from grav1synth_amd.estimate import format_estimates

    assert format_estimates([]) == b"filmgrn1\n"
    out = format_estimates([1.0, None, 2.34567, 0.0005, 12345.6785])
    lines = out.decode().split("\n")
    assert lines[:4] == ["filmgrn1", "1.000", "-1.000", "2.346"] and lines[-1] == ""
    # "{:.3}" prints the correctly rounded decimal of the f64, as printf does
    assert lines[4] == "%.3f" % 0.0005 and lines[5] == "%.3f" % 12345.6785


def test_estimate_front_door_refusals(tmp_path, caplog):
    from grav1synth_amd import cli

    a, out = str(tmp_path / "a.y4m"), str(tmp_path / "o.txt")
    with caplog.at_level(logging.INFO, logger="grav1synth"):
        assert cli.estimate_command(a, a) == -1
        assert caplog.records[-1].getMessage() == cli.SAME_AS_OUTPUT
        open(out, "w").write("keep me")
        assert cli.estimate_command(a, out, confirm=lambda p: False) == -1
        assert caplog.records[-1].getMessage() == cli.NOT_OVERWRITING
        assert open(out).read() == "keep me"


GPU_CASES = [
    (SynthSpec(320, 192, 8), "dev"), (SynthSpec(326, 198, 10), "dev"), (SynthSpec(501, 67, 12), "dev"),   # 501: the second column strip is 5 wide
    (SynthSpec(1920, 1080, 10), "dev"), (SynthSpec(3840, 2160, 10), "dev"), (SynthSpec(322, 190, 8), "host"),
    (SynthSpec(322, 190, 10), "strided"), (SynthSpec(17, 9, 8), "host"), (SynthSpec(3, 3, 10), "dev"), (SynthSpec(2, 40, 8), "host"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("bd", [8, 10, 12, 14, 16])
def test_hip_estimator_at_the_extremes_of_the_sample_range(bd, monkeypatch):
    """Full-range content: the packed 16-bit kernel (depths <= 12) holds |Gx| + |Gy| + rounding <= 32768 and |Laplacian| <= 8 * 4095
    in 16 bits; deeper samples take the 32-bit kernel.  Uniform noise over the whole range, a 0 / max checkerboard, isolated
    max samples on zero, a field one step under the edge threshold -- against the oracle, and packed against 32-bit."""
    import torch

    from grav1synth_amd.estimate import NoiseEstimator

    rng = np.random.default_rng(bd)
    dt = np.uint8 if bd == 8 else np.uint16
    mx = (1 << bd) - 1
    H, W = 70, 504
    planes = [rng.integers(0, mx + 1, (H, W)).astype(dt),
              ((np.indices((H, W)).sum(0) & 1) * mx).astype(dt),
              (rng.random((H, W)) < 0.02).astype(dt) * dt(mx),
              (rng.integers(0, 2, (H, W)) * (5 << (bd - 8))).astype(dt),
              np.full((H, W), mx, dt)]
    want = [estimate_plane_noise(p, bd) for p in planes]
    for mode in ("", "wide"):
        if mode:
            monkeypatch.setenv("G1S_ESTIMATE", mode)
        est = NoiseEstimator(bd, batch_frames=8)
        keep = [torch.from_numpy(p).cuda() for p in planes]
        for t in keep:
            est.estimate_frame(t)
        got = est.finish()
        est.close()
        assert got == want, mode


@pytest.mark.gpu
@pytest.mark.parametrize("spec,where", GPU_CASES, ids=lambda x: x if isinstance(x, str) else f"{x.width}x{x.height}_{x.bit_depth}b")
def test_hip_estimates_equal_the_oracle(spec, where):
    import torch

    from grav1synth_amd.estimate import NoiseEstimator, format_estimates

    est = NoiseEstimator(spec.bit_depth, batch_frames=3)
    want, keep = [], []
    for k in range(5):
        p = _plane(spec, k)
        if k == 3:
            p = np.full_like(p, 77)            # noise-free: 0.0
        if k == 4 and spec.width >= 8:
            p = ((np.indices(p.shape)[1] // 2 % 2) * ((1 << spec.bit_depth) - 1)).astype(p.dtype)  # edges everywhere: None
        want.append(estimate_plane_noise(p, spec.bit_depth))
        if where == "host":
            est.estimate_frame(p)
        elif where == "strided":               # a view into a wider device plane: odd pointer, pitch != width
            big = torch.zeros((p.shape[0] + 2, p.shape[1] + 7), dtype=torch.from_numpy(p).dtype, device="cuda")
            big[1:-1, 3:3 + p.shape[1]] = torch.from_numpy(p).cuda()
            keep.append(big)
            est.estimate_frame(big[1:-1, 3:3 + p.shape[1]])
        else:
            t = torch.from_numpy(p).cuda()
            keep.append(t)
            est.estimate_frame(t)
    got = est.finish()
    assert got == want   # bit for bit (floats compared exactly; None where the reference has None)
    if spec.width >= 8:
        assert got[3] == 0.0 and got[4] is None
    assert format_estimates(got).decode().splitlines()[0] == "filmgrn1"
    est.close()


@pytest.mark.gpu
def test_estimate_command_on_a_y4m_file(tmp_path):
    from grav1synth_amd import cli
    from grav1synth_amd.ingest import write_y4m

    spec = SynthSpec(320, 200, 10)
    frames, want = [], ["filmgrn1"]
    for k in range(4):
        s, _ = make_pair(spec, k)
        planes = [p.numpy() for p in s]
        frames.append(planes)
        e = estimate_plane_noise(planes[0], 10)
        want.append("%.3f" % (-1.0 if e is None else e))
    src, out = str(tmp_path / "s.y4m"), str(tmp_path / "o.txt")
    write_y4m(src, frames, 10, 1, 1)
    assert cli.main(["estimate", src, "-o", out, "-y"]) == 0
    assert open(out).read().splitlines() == want
