"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle.

Bit-exact bar: flat mask bytes, f32 score bits, every integer AR sum, every
per-block statistic and the final `.tbl` bytes must equal the oracle's.
Mirrors how the reference would test `diff` if it had tests: construct the
generator (src/main.rs:420-427), feed frame pairs (:442), finish (:524), write
the table (:525-529).
"""
from fractions import Fraction

import numpy as np
import pytest
import torch

from grav1synth_amd.diff import DiffGenerator, Frame, format_tbl
from grav1synth_amd.synth import SynthSpec, make_pair
from tests.helpers import np_pair, oracle_run

pytestmark = pytest.mark.gpu

CASES = [
    # (spec, lag, chroma, nframes, on_device)
    (SynthSpec(320, 192, 8), 3, True, 3, True),
    (SynthSpec(320, 200, 8), 3, True, 2, False),           # partial bottom block row (200 = 6.25 blocks)
    (SynthSpec(352, 208, 10), 3, True, 2, True),           # 10-bit u16, 4:2:0
    (SynthSpec(320, 192, 10, xdec=1, ydec=0), 3, True, 2, True),   # 4:2:2
    (SynthSpec(256, 160, 10, xdec=0, ydec=0), 3, True, 2, True),   # 4:4:4
    (SynthSpec(320, 192, 8), 2, False, 2, True),           # BASELINE config[1]: lag 2, luma only
    (SynthSpec(320, 192, 8), 1, True, 2, True),
    (SynthSpec(300, 180, 8, textured=False), 3, True, 2, False),  # width not a multiple of 32
    (SynthSpec(320, 192, 12), 3, True, 2, True),
    (SynthSpec(326, 198, 8), 3, True, 2, True),            # nothing a multiple of 4 or 8; unaligned device rows
    (SynthSpec(322, 190, 10), 3, True, 2, False),          # same through the host staging path, 10-bit
    (SynthSpec(64, 64, 8), 3, True, 2, True),              # 2 x 2 blocks: every area touches the frame edge
    (SynthSpec(512, 40, 8, textured=False), 3, True, 2, True),   # a 1.25-block-high strip
    (SynthSpec(288, 160, 8, xdec=0, ydec=0), 3, True, 2, True),  # 4:4:4 8-bit
    # lag 1 / 2 run through the lag-3 tiles with their own (narrower) window borders
    (SynthSpec(326, 198, 8), 2, True, 2, True),            # odd sizes, chroma
    (SynthSpec(320, 200, 10), 1, True, 2, False),          # partial bottom block row, host frames
    (SynthSpec(256, 160, 10, xdec=0, ydec=0), 2, True, 2, True),   # 4:4:4
    (SynthSpec(320, 192, 10, xdec=1, ydec=0), 1, False, 2, True),  # 4:2:2 source, luma only
    (SynthSpec(64, 64, 8), 2, True, 2, True),              # every area touches the frame edge
    (SynthSpec(3840, 2160, 10), 3, True, 1, True),         # the bench workload at full size, one frame pair
    (SynthSpec(320, 192, 8, xdec=0, ydec=1), 3, True, 2, True),   # 4:4:0 (not a format the reference reads): generic kernel
]


def _ids(c):
    s = c[0]
    return f"{s.width}x{s.height}_{s.bit_depth}b_{s.xdec}{s.ydec}_lag{c[1]}_{'yuv' if c[2] else 'y'}_{'dev' if c[4] else 'host'}"


@pytest.mark.parametrize("case", CASES, ids=_ids)
def test_records_and_table_match_oracle(case):
    spec, lag, chroma, nframes, on_device = case
    g = DiffGenerator(Fraction(24, 1), spec.bit_depth, spec.bit_depth, ar_coeff_lag=lag, luma_only=not chroma,
                      batch_frames=1)
    nplanes = 3 if chroma else 1
    mismatches = []

    def collect(o, k):
        if on_device:
            s, d = make_pair(spec, k, device="cuda")
        else:
            s, d = np_pair(spec, k)
        g.diff_frame(Frame(s, spec.xdec, spec.ydec), Frame(d, spec.xdec, spec.ydec))
        g.sync()
        r = g.last_record()
        om, rm = o.flat_mask(), r.flat_mask()
        if not np.array_equal(om, rm):
            mismatches.append(f"frame {k}: flat mask differs at {np.argwhere(om != rm)[:5].tolist()}")
        osc, rsc = o.scores(), r.scores()
        if not np.array_equal(osc.view(np.uint32), rsc.view(np.uint32)):
            mismatches.append(f"frame {k}: score bits differ ({(osc.view(np.uint32) != rsc.view(np.uint32)).sum()} blocks)")
        flat = om.ravel() != 0
        for c in range(nplanes):
            S, Sb, nobs = o.ar_sums(c)
            S2, Sb2, nobs2 = r.ar_sums(c)
            if nobs != nobs2 or not np.array_equal(S, S2) or not np.array_equal(Sb, Sb2):
                mismatches.append(f"frame {k} plane {c}: AR sums differ (nobs {nobs} vs {nobs2})")
            ls, sd, sd2 = o.block_stats(c)
            ls2, sd_2, sd2_2 = r.block_stats(c)
            # the oracle records statistics only for blocks it measures (flat, > 32 samples)
            # (per plane: a chroma corner block of <= 32 samples is skipped while its luma block is measured)
            meas = flat & ((sd2 != 0) | (sd != 0) | ((ls != 0) if c == 0 else False))
            if c == 0 and not np.array_equal(ls[meas], ls2[meas]):
                mismatches.append(f"frame {k}: luma block sums differ")
            if not np.array_equal(sd[meas], sd_2[meas]) or not np.array_equal(sd2[meas], sd2_2[meas]):
                mismatches.append(f"frame {k} plane {c}: block noise sums differ")

    tbl, _ = oracle_run(spec, range(nframes), lag, chroma, collect=collect)
    mine = format_tbl(g.finish())
    assert not mismatches, "\n".join(mismatches)
    assert mine == tbl


@pytest.mark.parametrize("nbatches,no_defer", [(5, False), (7, False), (9, False), (5, True), (8, False)])
def test_two_ranks_stream_a_sharded_job_on_one_device(tmp_path, nbatches, no_defer):
    """StreamingShardedDiff -- the class `bench.py --gpus N` runs -- with TWO ranks (gloo, both on this one GPU): batches
    dealt round-robin (an odd count: rank 1 sits out the last round and is one feed behind from then on -- with 7 and 9
    batches, and with 5 under G1S_NO_DEFER, the ranks send different local batches in the same round: the root orders by
    the batch index in the message), a scene cut inside a batch, the exchange of latest states, the ordered merge on
    rank 0.  The table must be the single generator's and the oracle's, byte for byte."""
    import os
    import socket
    import subprocess
    import sys

    from tests import dist_gpu_worker as w

    SPECS = w.specs(nbatches)

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "sharded.tbl")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   G1S_FOLD_THREADS="4", G1S_TEST_BATCHES=str(nbatches))
        if no_defer:
            env["G1S_NO_DEFER"] = "1"
        procs.append(subprocess.Popen([sys.executable, "-m", "tests.dist_gpu_worker", out], env=env, cwd=root,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(x[-1500:] for x in logs)
    want, segs = oracle_run(w.A, range(len(SPECS)), fps=w.FPS, specs_per_frame=SPECS)
    assert len(segs) >= 2
    g = DiffGenerator(w.FPS, 8, 8, batch_frames=w.BATCH)
    for k, sp in enumerate(SPECS):
        s, d = make_pair(sp, k, device="cuda")
        g.diff_frame(s, d, 1, 1)
    single = format_tbl(g.finish())
    assert single == want
    assert open(out, "rb").read() == want


@pytest.mark.parametrize("name", ["oracle_full_1920x1080_8b_lag2_luma.tbl", "oracle_full_1920x1080_8b_420_lag3.tbl",
                                  "oracle_full_7680x4320_10b_444_lag3.tbl", "oracle_full_1920x1080_8b_420_lag3_30frames.tbl",
                                  "oracle_full_3840x2160_10b_420_lag3_cut.tbl"])
def test_full_size_tables_match_the_committed_oracle_goldens(name):
    """BASELINE.json's configurations at their FULL sizes -- 1080p 8-bit lag 2 luma-only (configs[1]), 1080p 8-bit 4:2:0
    lag 3 (3 frames, and configs[0] as stated: 30 frames), 8K 10-bit 4:4:4 lag 3 (configs[4]'s format), 4K 10-bit 4:2:0 lag 3
    (configs[2]'s format) with a scene cut after four of eight frames -- against tables the oracle wrote for the same seeded
    frames (tests/golden/make_golden.py full: minutes of CPU time, committed as data): byte-identical."""
    import os

    from tests.golden import make_golden

    gd = make_golden.FULL_SIZE[name]
    spec, lag, chroma = gd["spec"], gd["lag"], gd["chroma"]
    with open(os.path.join(os.path.dirname(__file__), "golden", name), "rb") as f:
        want = f.read()
    fps = Fraction(*gd.get("fps", (24, 1))) if "cut" in gd else Fraction(24, 1)
    g = DiffGenerator(fps, spec.bit_depth, spec.bit_depth, ar_coeff_lag=lag, luma_only=not chroma, batch_frames=2 if gd["frames"] < 8 else 5)
    for k in range(gd["frames"]):
        sp = spec
        if "cut" in gd and k >= gd["cut"]:
            sp = SynthSpec(spec.width, spec.height, spec.bit_depth, gain_scale=3)
        s, d = make_pair(sp, k, device="cuda")
        if not chroma:
            s, d = s[:1], d[:1]
        g.diff_frame(Frame(s, spec.xdec, spec.ydec), Frame(d, spec.xdec, spec.ydec))
        del s, d
    got = format_tbl(g.finish())
    assert got == want
    if "cut" in gd:
        assert got.count(b"E ") == 2 if isinstance(got, bytes) else got.count("E ") == 2  # (the cut is there: two segments)


def _padded(planes, spec, top, bottom, left, right, fill):
    """the planes with a border around them (what the crop filter takes off again)"""
    out = []
    for c, p in enumerate(planes):
        sx, sy = (spec.xdec, spec.ydec) if c else (0, 0)
        t, b, l, r = top >> sy, bottom >> sy, left >> sx, right >> sx
        q = np.full((p.shape[0] + t + b, p.shape[1] + l + r), fill + c, dtype=p.dtype)
        q[t:t + p.shape[0], l:l + p.shape[1]] = p
        out.append(q)
    return out


@pytest.mark.parametrize("bd", [8, 10])
def test_front_door_with_a_cropped_source_equals_the_oracle_on_the_cropped_planes(tmp_path, bd, caplog):
    """N3 + the front door: `diff SOURCE DENOISED -o OUT -y -f crop:...` on .y4m files whose source carries a border
    (letterbox) the denoised file does not: the table must be, byte for byte, the oracle's table for the bare planes --
    the crop is extent arithmetic in the frame descriptors, the kernels see a smaller frame with the same strides."""
    import logging

    from grav1synth_amd import cli
    from grav1synth_amd.ingest import write_y4m

    spec = SynthSpec(320, 200, bd)
    top, bottom, left, right = 6, 2, 16, 4
    nframes = 3
    src, den = [], []
    for k in range(nframes):
        s, d = np_pair(spec, k)
        src.append(_padded(s, spec, top, bottom, left, right, 37))
        den.append(d)
    fps = Fraction(30000, 1001)
    a, b, out = str(tmp_path / "source.y4m"), str(tmp_path / "denoised.y4m"), str(tmp_path / "out.tbl")
    write_y4m(a, src, bd, spec.xdec, spec.ydec, fps)
    write_y4m(b, den, bd, spec.xdec, spec.ydec, fps)
    want, _ = oracle_run(spec, range(nframes), fps=fps)
    open(out, "w").write("stale")
    with caplog.at_level(logging.INFO):
        n = cli.diff_command(a, b, out, overwrite=True, filters=f"crop:top={top},bottom={bottom},left={left},right={right}")
    assert n == nframes
    assert open(out, "rb").read() == want
    msgs = [r.getMessage() for r in caplog.records]
    assert f"Computed diff for {nframes} frames" in msgs and f"Done, wrote output file to {out}" in msgs  # src/main.rs:531-532
    # without the filter the geometry differs: the reference's error, with the frame pair it happened on
    from grav1synth_amd.ingest import diff_y4m_files
    with pytest.raises(RuntimeError, match=r"frame 0: .*dimensions do not match"):
        diff_y4m_files(a, b, out)


@pytest.mark.parametrize("alg", ["hermite", "catmullrom", "mitchell", "lanczos", "spline36"])
def test_device_resize_equals_the_oracle_plane_for_plane(alg):
    """N3: the resize filter on the device (csrc/resize.hip) against oracle/resize_oracle.c, bit for bit: up and down, odd
    sizes whose windows mirror at the edges, 8- and 10-bit, 4:2:0 and 4:4:4 chroma planes, host and device input."""
    from grav1synth_amd.filters import FilterChain
    from tests.oracle_binding import resize_planes

    rng = np.random.default_rng(11)
    for (w, h, bd, xd, yd, tw, th, on_dev) in [(96, 64, 8, 1, 1, 48, 40, False), (100, 38, 10, 1, 1, 146, 90, False), (64, 48, 10, 0, 0, 37, 29, True),
                                               (90, 70, 8, 0, 0, 90, 70, True), (320, 200, 10, 1, 1, 480, 304, True)]:
        dt = np.uint8 if bd == 8 else np.uint16
        planes = [rng.integers(0, 1 << bd, (h >> (yd if c else 0), w >> (xd if c else 0)), dtype=dt) for c in range(3)]
        planes[0][: h // 4] = (1 << bd) - 1  # (saturated and empty areas: the clamp)
        planes[0][h // 4: h // 2, : w // 3] = 0
        want = resize_planes(planes, xd, yd, tw, th, bd, alg)
        src = [torch.from_numpy(p).cuda() for p in planes] if on_dev else planes
        got = FilterChain(f"resize:width={tw},height={th},alg={alg}").apply(Frame(src, xd, yd), bd).planes
        for c in range(3):
            assert got[c].shape == want[c].shape and np.array_equal(got[c], want[c]), f"{w}x{h} -> {tw}x{th} {bd}b plane {c}"


@pytest.mark.parametrize("bd,alg", [(8, "catmullrom"), (10, "lanczos")])
def test_front_door_with_a_resized_source_equals_the_oracle_on_the_oracles_resized_planes(tmp_path, bd, alg):
    """N3 + the front door: `diff SOURCE DENOISED -o OUT -y -f resize:width=..,height=..,alg=..` (src/filters.rs:150-178 applied
    to the source only, src/main.rs:615-629) on a source file of another size: the table is, byte for byte, the oracle's table
    for the source planes resized by the oracle's restatement; and crop and resize in one chain."""
    from grav1synth_amd import cli
    from grav1synth_amd.ingest import write_y4m
    from tests.oracle_binding import OracleDiff, format_tbl as ofmt, resize_planes

    spec = SynthSpec(320, 200, bd)
    nframes = 3
    fps = Fraction(30000, 1001)
    big, den = [], []
    for k in range(nframes):
        s, d = np_pair(spec, k)
        big.append(resize_planes(s, 1, 1, 480, 304, bd, "spline36"))  # (a larger source: 480 x 304)
        den.append(d)
    a, b, out = str(tmp_path / "source.y4m"), str(tmp_path / "denoised.y4m"), str(tmp_path / "out.tbl")
    write_y4m(a, big, bd, 1, 1, fps)
    write_y4m(b, den, bd, 1, 1, fps)
    o = OracleDiff(fps.numerator, fps.denominator, bd, bd, 3, True)
    for k in range(nframes):
        o.diff_frame(resize_planes(big[k], 1, 1, 320, 200, bd, alg), den[k], 1, 1)
    want = ofmt(o.finish())
    assert cli.diff_command(a, b, out, overwrite=True, filters=f"resize:width=320,height=200,alg={alg}") == nframes
    assert open(out, "rb").read() == want
    # crop first, then resize what is left
    o = OracleDiff(fps.numerator, fps.denominator, bd, bd, 3, True)
    for k in range(nframes):
        cropped = [p[(8 >> (1 if c else 0)):, (16 >> (1 if c else 0)):] for c, p in enumerate(big[k])]
        o.diff_frame(resize_planes(cropped, 1, 1, 320, 200, bd, alg), den[k], 1, 1)
    assert cli.diff_command(a, b, out, overwrite=True, filters=f"crop:top=8,left=16;resize:width=320,height=200,alg={alg}") == nframes
    assert open(out, "rb").read() == ofmt(o.finish())


def test_cropped_device_frames_equal_the_oracle():
    """FilterChain.apply on device-resident planes (torch views: pointer + extent arithmetic, unaligned rows)."""
    from grav1synth_amd.filters import FilterChain

    spec = SynthSpec(326, 198, 10)
    chain = FilterChain("crop:top=2,left=6;crop:bottom=4,right=2")
    want, _ = oracle_run(spec, range(2))
    g = DiffGenerator(Fraction(24, 1), 10, 10, batch_frames=2)
    keep = []
    for k in range(2):
        s, d = np_pair(spec, k)
        big = [torch.from_numpy(p).cuda() for p in _padded(s, spec, 2, 4, 6, 2, 900)]
        dd = [torch.from_numpy(p).cuda() for p in d]
        keep.append((big, dd))
        g.diff_frame(chain.apply(Frame(big, spec.xdec, spec.ydec)), Frame(dd, spec.xdec, spec.ydec))
    assert format_tbl(g.finish()) == want


def test_pinned_host_frames_are_copied_asynchronously_and_give_the_same_table():
    """N2: on_device = 2 (pinned host planes, Frame(..., async_host=True)): g1s_diff_frame queues the copies and returns;
    g1s_diff_frames_copied says when the planes may be reused; the table is the oracle's."""
    spec = SynthSpec(322, 190, 10)
    want, _ = oracle_run(spec, range(5))
    g = DiffGenerator(Fraction(24, 1), 10, 10, batch_frames=2)
    for k in range(5):
        s, d = np_pair(spec, k)
        ps = [torch.from_numpy(p).pin_memory() for p in s]
        pd = [torch.from_numpy(p).pin_memory() for p in d]
        f = Frame(ps, spec.xdec, spec.ydec, async_host=True)
        assert f.to_c([]).on_device == 2
        g.diff_frame(f, Frame(pd, spec.xdec, spec.ydec, async_host=True))
        assert g.frames_copied() <= k + 1
        assert g.frames_copied(k + 1) == k + 1  # wait for this frame: now its planes may be overwritten
        for p in ps + pd:
            p.fill_(0)
    assert format_tbl(g.finish()) == want


def test_pinned_staging_buffers_can_be_refilled_right_after_diff_frame():
    """a pinned tensor is usually a staging buffer its owner refills; without async_host the planes are
    copied before diff_frame returns (the `&Frame` borrow), so ONE pair of pinned buffers can carry every frame."""
    spec = SynthSpec(322, 190, 10)
    want, _ = oracle_run(spec, range(5))
    g = DiffGenerator(Fraction(24, 1), 10, 10, batch_frames=2)
    s0, d0 = np_pair(spec, 0)
    ps = [torch.from_numpy(p.copy()).pin_memory() for p in s0]
    pd = [torch.from_numpy(p.copy()).pin_memory() for p in d0]
    assert Frame(ps, spec.xdec, spec.ydec).to_c([]).on_device == 0
    with pytest.raises(ValueError):
        Frame([p.numpy() for p in ps], spec.xdec, spec.ydec, async_host=True).to_c([])
    for k in range(5):
        s, d = np_pair(spec, k)
        for t, p in zip(ps + pd, list(s) + list(d)):
            t.copy_(torch.from_numpy(p))
        g.diff_frame(Frame(ps, spec.xdec, spec.ydec), Frame(pd, spec.xdec, spec.ydec))
        for t in ps + pd:
            t.fill_(0)
    assert format_tbl(g.finish()) == want


def test_truncated_y4m_reports_the_frame_index(tmp_path):
    """N2: an error of the frame-pair loop names the frame pair (g1s_diff_run_filtered)."""
    from grav1synth_amd.ingest import diff_y4m_files, write_y4m

    spec = SynthSpec(320, 192, 8)
    src, den = [], []
    for k in range(4):
        s, d = np_pair(spec, k)
        src.append(s)
        den.append(d)
    a, b = str(tmp_path / "a.y4m"), str(tmp_path / "b.y4m")
    write_y4m(a, src, 8, 1, 1)
    write_y4m(b, den, 8, 1, 1)
    size = __import__("os").path.getsize(b)
    with open(b, "r+b") as f:
        f.truncate(size - 1000)  # the 4th denoised frame is short
    with pytest.raises(RuntimeError, match=r"frame 3: denoised reader failed"):
        diff_y4m_files(a, b, str(tmp_path / "o.tbl"))


def test_hip_table_drives_the_apply_lookup():
    """N1 closure: the table the HIP path emits for a two-segment job, read back with g1s_parse_tbl and walked with
    g1s_tbl_segment_for over the frames' presentation times, stamps every frame with the right segment and advances that
    segment's seed by DEFAULT_GRAIN_SEED per hit (wrapping u16) -- `apply`'s per-frame lookup,
    /root/reference/src/parser/frame.rs:617-633 (its test :4590-4608: seed 100 -> 100 + DEFAULT_GRAIN_SEED)."""
    from grav1synth_amd.tbl import GrainTable, parse_tbl_native

    DEFAULT_GRAIN_SEED = 10956
    a = SynthSpec(320, 192, 8)
    b = SynthSpec(320, 192, 8, gain_scale=3)
    specs = [a, a, a, b, b, b]
    fps = Fraction(30000, 1001)
    g = DiffGenerator(fps, 8, 8, batch_frames=4)
    for k, sp in enumerate(specs):
        s, d = make_pair(sp, k, device="cuda")
        g.diff_frame(s, d, sp.xdec, sp.ydec)
    out = g.finish()
    assert len(out) >= 2
    text = format_tbl(out)
    segs = parse_tbl_native(text)
    assert [(x.start_time, x.end_time, x.random_seed) for x in segs] == [(x.start_time, x.end_time, x.random_seed) for x in out]
    table = GrainTable(segs)
    hits = [0] * len(segs)
    assert segs[0].end_time == (3 * 10_000_000 * fps.denominator) // fps.numerator, "the first cut sits at frame 3"
    for k in range(len(specs)):
        ts = (k * 10_000_000 * fps.denominator) // fps.numerator  # the frame's presentation time in the table's 10 MHz ticks
        want = next(i for i, x in enumerate(segs) if x.start_time <= ts < x.end_time)
        assert (want == 0) == (k < 3)
        seg = table.segment_for(ts)
        assert seg is not None and seg.start_time == segs[want].start_time
        hits[want] += 1
        assert seg.random_seed == (segs[want].random_seed + hits[want] * DEFAULT_GRAIN_SEED) & 0xFFFF
        assert seg.scaling_points_y == segs[want].scaling_points_y and seg.ar_coeffs_y == segs[want].ar_coeffs_y
    assert sum(hits) == len(specs) and hits[0] == 3
    assert table.segment_for(segs[-1].end_time) is None  # past the end: `apply` leaves the frame alone


def test_streamed_device_frames_are_released_batch_by_batch():
    """A long GPU-resident stream must not pin every frame until finish: g1s_diff_frames_released tells the caller
    which frame pairs the generator is done reading, and the Python mirror prunes its keep-alives by it."""
    import time

    spec = SynthSpec(320, 192, 8)
    g = DiffGenerator(Fraction(24, 1), 8, 8, batch_frames=4)
    n, worst = 64, 0
    for k in range(n):
        s, d = make_pair(spec, k % 5, device="cuda")
        g.diff_frame(s, d, spec.xdec, spec.ydec)
        del s, d
        if k % 8 == 7:
            time.sleep(0.02)  # (let the pipeline drain a little: a real producer is slower than these tiny frames)
        worst = max(worst, len(g._keep))
        assert g.frames_released() <= k + 1
    assert worst <= 4 * 4 + 8, f"{worst} frame pairs pinned at once"  # four slots of four frames + what the sleep covers
    g.sync()
    assert g.frames_released() == n
    assert len(g.finish()) >= 1


def test_batched_equals_unbatched_and_oracle():
    """Batching/pipelining must not change anything: 7 frames with batch 3."""
    spec = SynthSpec(320, 192, 8)
    tbl, _ = oracle_run(spec, range(7))
    g = DiffGenerator(Fraction(24, 1), 8, 8, batch_frames=3)
    for k in range(7):
        s, d = make_pair(spec, k, device="cuda")
        g.diff_frame(s, d, spec.xdec, spec.ydec)
    assert format_tbl(g.finish()) == tbl


def test_full_size_batching_and_pipeline_depth_do_not_change_the_records():
    """4K 10-bit 4:2:0 (BASELINE configs[2]): the per-frame records must not depend on how frames are grouped
    into launches (1, 3 or 5 per batch: the last batches ragged), nor on where in the four-slot pipeline a
    frame sits -- a size-independent property checked at the full size."""
    spec = SynthSpec(3840, 2160, 10)
    pairs = [make_pair(spec, k, device="cuda") for k in range(7)]
    ref = None
    for bf in (1, 3, 5):
        g = DiffGenerator(Fraction(24, 1), 10, 10, batch_frames=bf, records_only=True)
        for s, d in pairs:
            g.diff_frame(s, d, 1, 1)
        recs, n = g.take_records(spec.width, spec.height, 3, len(pairs))
        g.close()
        assert n == len(pairs)
        if ref is None:
            ref = recs.copy()
        else:
            assert np.array_equal(ref, recs), f"batch_frames={bf}"


def test_latest_only_generator_streams_the_fold():
    """Frame-shard streaming mode (records_only = 2): the generator keeps per-frame latest states; taking
    them batch by batch and merging them in order gives the single-GPU table, byte for byte."""
    from grav1synth_amd.diff import RecordFold, latest_size

    spec = SynthSpec(320, 192, 8)
    tbl, _ = oracle_run(spec, range(7))
    g = DiffGenerator(Fraction(24, 1), 8, 8, batch_frames=3, records_only=2)
    fold = RecordFold(Fraction(24, 1), 3)
    seen = 0
    for k in range(7):
        s, d = make_pair(spec, k, device="cuda")
        g.diff_frame(s, d, spec.xdec, spec.ydec)
        blobs = g.take_latest(16)  # complete batches only, no waiting
        assert blobs.shape[1] == latest_size(3)
        seen += blobs.shape[0]
        fold.push_latest_many(blobs)
    blobs = g.take_latest(16, sync=True)
    seen += blobs.shape[0]
    fold.push_latest_many(blobs)
    assert seen == 7
    assert format_tbl(fold.finish()) == tbl
    g.close()


@pytest.mark.parametrize("bd,w,h", [(8, 1920, 1080), (10, 2048, 1152), (12, 1280, 736)])
def test_certified_flat_finder_equals_literal_kernel(bd, w, h):
    """The flat-block finder's fast path (integer moments + certified evaluation, literal kernel for the
    blocks it cannot decide) and the wave-per-block literal kernel must give the lane-per-block literal kernel's mask bytes and f32 score bits for EVERY
    block: tens of thousands of blocks here, textured and flat, on top of the oracle cases above."""
    spec = SynthSpec(w, h, bd)
    res = []
    for mode in (0, 1, 2):
        g = DiffGenerator(Fraction(24, 1), bd, bd, batch_frames=4)
        g.set_flat_finder(mode)
        g.set_timing(True)
        masks, scores = [], []
        for k in range(4):
            s, d = make_pair(spec, 100 + k, device="cuda")
            g.diff_frame(s, d, spec.xdec, spec.ydec)
            g.sync()
            r = g.last_record()
            masks.append(r.flat_mask().copy())
            scores.append(r.scores().view(np.uint32).copy())
        st = g.stats()
        res.append((masks, scores, st.literal_blocks, st.blocks))
        g.close()
    (m0, s0, lit0, nb0), (m1, s1, lit1, nb1), (m2, s2, lit2, nb2) = res
    for other_m, other_s in ((m0, s0), (m2, s2)):
        for a, b in zip(other_m, m1):
            assert np.array_equal(a, b)
        for a, b in zip(other_s, s1):
            assert np.array_equal(a, b)
    assert lit1 == nb1 and lit2 == nb2   # literal modes: every block through a literal kernel
    assert lit0 < nb0 // 100             # fast path: only the undecidable few


def test_scene_cut_emits_two_segments():
    """is_different(): doubling the noise gain mid-stream must cut a segment at
    the same frame, with the same timestamps, as the oracle."""
    a = SynthSpec(320, 192, 8)
    b = SynthSpec(320, 192, 8, gain_scale=3)
    specs = [a, a, a, b, b, b]
    fps = Fraction(30000, 1001)
    tbl, segs = oracle_run(a, range(6), specs_per_frame=specs, fps=fps)
    assert len(segs) >= 2
    g = DiffGenerator(fps, 8, 8, batch_frames=4)
    for k, sp in enumerate(specs):
        s, d = make_pair(sp, k, device="cuda")
        g.diff_frame(s, d, sp.xdec, sp.ydec)
    out = g.finish()
    assert format_tbl(out) == tbl
    assert out[0].random_seed == 10956 and out[1].random_seed == 0


def test_mixed_bit_depths():
    """(8, 9..=16) and (9..=16, 8) monomorphisations of src/main.rs:434-518."""
    s8 = SynthSpec(320, 192, 8)
    s10 = SynthSpec(320, 192, 10)
    from tests.oracle_binding import OracleDiff, format_tbl as ofmt

    for sb, db in ((8, 10), (10, 8)):
        o = OracleDiff(24, 1, sb, db, 3, True)
        g = DiffGenerator(Fraction(24, 1), sb, db)
        for k in range(2):
            src = np_pair(s8 if sb == 8 else s10, k)[0]
            den = np_pair(s8 if db == 8 else s10, k)[1]
            o.diff_frame(src, den, 1, 1)
            g.diff_frame([torch.from_numpy(p).cuda() for p in src], [torch.from_numpy(p).cuda() for p in den])
        assert format_tbl(g.finish()) == ofmt(o.finish())


@pytest.mark.parametrize("sb,db,chroma,size", [(10, 8, True, (352, 208)), (8, 10, True, (352, 208)), (10, 12, True, (320, 200)), (12, 10, False, (384, 192)),
                                              (10, 8, False, (320, 192)), (16, 10, True, (320, 192))])
def test_mixed_depths_records_match_oracle(sb, db, chroma, size):
    """Source and denoised frames of different depths (4:2:0 and luma-only: the wide chain's general residual form -- each input
    narrowed on its own, engine.hip wide_gen): every frame's mask, score bits, integer sums and block statistics, and the table,
    against the oracle; block rows cut at the bottom (200 = 6.25 blocks)."""
    from tests.oracle_binding import OracleDiff, format_tbl as ofmt

    w, h = size
    ss, sd = SynthSpec(w, h, sb), SynthSpec(w, h, db)
    o = OracleDiff(24, 1, sb, db, 3, chroma)
    g = DiffGenerator(Fraction(24, 1), sb, db, luma_only=not chroma, batch_frames=1)
    nplanes = 3 if chroma else 1
    for k in range(3):
        src, den = np_pair(ss, k)[0], np_pair(sd, k)[1]
        if k == 2:  # (a few residuals outside int8: the deferred path of the general form)
            den = [p.copy() for p in den]
            den[0][40:43, 100:104] = 0
        o.diff_frame(src, den, 1, 1)
        g.diff_frame(Frame([torch.from_numpy(p).cuda() for p in src], 1, 1), Frame([torch.from_numpy(p).cuda() for p in den], 1, 1))
        g.sync()
        r = g.last_record()
        assert np.array_equal(o.flat_mask(), r.flat_mask()), f"frame {k}: flat mask"
        assert np.array_equal(o.scores().view(np.uint32), r.scores().view(np.uint32)), f"frame {k}: score bits"
        flat = o.flat_mask().ravel() != 0
        for c in range(nplanes):
            S, Sb, nobs = o.ar_sums(c)
            S2, Sb2, nobs2 = r.ar_sums(c)
            assert nobs == nobs2 and np.array_equal(S, S2) and np.array_equal(Sb, Sb2), f"frame {k} plane {c}: AR sums"
            ls, sdd, sd2 = o.block_stats(c)
            ls2, sdd2, sd22 = r.block_stats(c)
            meas = flat & ((sd2 != 0) | (sdd != 0) | ((ls != 0) if c == 0 else False))
            if c == 0:
                assert np.array_equal(ls[meas], ls2[meas]), f"frame {k}: luma block sums"
            assert np.array_equal(sdd[meas], sdd2[meas]) and np.array_equal(sd2[meas], sd22[meas]), f"frame {k} plane {c}: block noise sums"
    assert format_tbl(g.finish()) == ofmt(o.finish())


def test_errors_match_reference_behaviour():
    from grav1synth_amd._lib import G1SError

    # dimension mismatch -> error from diff_frame (anyhow::Error at src/main.rs:442)
    g = DiffGenerator(Fraction(24, 1), 8, 8)
    a = np.zeros((64, 64), np.uint8)
    b = np.zeros((64, 96), np.uint8)
    with pytest.raises(G1SError) as e:
        g.diff_frame([a], [b])
    assert e.value.code == -2
    # a constant frame has no flat blocks worth using -> "Not enough flat blocks"
    g = DiffGenerator(Fraction(24, 1), 8, 8, batch_frames=1)
    z = [np.full((64, 64), 7, np.uint8), np.full((32, 32), 7, np.uint8), np.full((32, 32), 7, np.uint8)]
    # 64x64 = 4 blocks, all-zero scores: threshold 0 -> every block flagged |= 1 -> solvable? the
    # luma AR system is all zeros -> singular -> solve error, as in the oracle
    from tests.oracle_binding import OracleDiff

    o = OracleDiff(24, 1, 8, 8, 3, True)
    with pytest.raises(RuntimeError):
        o.diff_frame(z, z, 1, 1)
    with pytest.raises(G1SError):
        g.diff_frame(z, z)
        g.sync()
    # finish() consumes the generator
    g2 = DiffGenerator(Fraction(24, 1), 8, 8)
    spec = SynthSpec(128, 96, 8, textured=False)
    s, d = np_pair(spec, 0)
    g2.diff_frame(s, d)
    g2.finish()
    with pytest.raises(G1SError) as e:
        g2.diff_frame(s, d)
    assert e.value.code == -7


@pytest.mark.parametrize("bd,xdec,ydec,lag", [(8, 1, 1, 3), (10, 0, 0, 3), (8, 1, 1, 2), (10, 1, 0, 1)])
def test_large_residuals_take_the_deferred_path(bd, xdec, ydec, lag):
    """|src - den| > 127 does not fit the int8 dot4 path: those blocks must be
    handled by the generic kernel with identical integer sums."""
    spec = SynthSpec(320, 192, bd, xdec=xdec, ydec=ydec)
    up = bd - 8
    o_args = (24, 1, bd, bd, lag, True)
    from tests.oracle_binding import OracleDiff, format_tbl as ofmt

    o = OracleDiff(*o_args)
    g = DiffGenerator(Fraction(24, 1), bd, bd, batch_frames=2, ar_coeff_lag=lag)
    rng = np.random.default_rng(7)
    for k in range(2):
        s, d = np_pair(spec, k)
        d = [p.copy() for p in d]
        # damage the DENOISED side only (the flat-block finder looks at the source)
        for c in range(3):
            h, w = d[c].shape
            for _ in range(12):
                y, x = int(rng.integers(0, h)), int(rng.integers(0, w * 2 // 3))
                d[c][y, x] = 0 if s[c][y, x] > (140 << up) else (255 << up)
        o.diff_frame(s, d, xdec, ydec)
        g.diff_frame([torch.from_numpy(p).cuda() for p in s], [torch.from_numpy(p).cuda() for p in d], xdec, ydec)
        if k == 1:
            g.sync()
            r = g.last_record()
            for c in range(3):
                S, Sb, n = o.ar_sums(c)
                S2, Sb2, n2 = r.ar_sums(c)
                assert n == n2 and np.array_equal(S, S2) and np.array_equal(Sb, Sb2), f"plane {c}"
    assert format_tbl(g.finish()) == ofmt(o.finish())


def test_halo_dwords_from_neighbouring_units_change_no_sum(monkeypatch):
    """The halo dwords of a unit come out of the registers of the units next to it in the workgroup's run (a ghost unit at
    either end of the run), never from memory: few workgroups per frame and many (every workgroup walks many units; slices
    that begin and end inside a block row), a few out-of-int8 residuals in last / first words of units (the flags a unit
    passes to its neighbours) -- the same integer sums and the same table either way, and the oracle's.  (G1S_F_*: the same
    switches of the stream chain, for the formats that run it.)"""
    spec = SynthSpec(1280, 352, 10, xdec=1, ydec=1, textured=False)
    frames = []
    rng = np.random.default_rng(5)
    for k in range(3):
        s, d = np_pair(spec, k)
        d = [p.copy() for p in d]
        for _ in range(6):  # the last word of a unit / the first word of the next one
            y = int(rng.integers(0, spec.height)); u = int(rng.integers(1, spec.width // 64 - 1))
            x = 64 * u - 1 - int(rng.integers(0, 3)) if rng.random() < 0.5 else 64 * u + int(rng.integers(0, 3))
            d[0][y, x] = 0 if (int(s[0][y, x]) >> 2) > 140 else (255 << 2)
        frames.append((s, d))
    out = {}
    for reuse in ("1", "0"):
        monkeypatch.setenv("G1S_F_REUSE", reuse)
        monkeypatch.setenv("G1S_F_WGS", "64")
        monkeypatch.setenv("G1S_W_WGS", "24" if reuse == "1" else "264")   # (3 frames: 8 / 88 workgroups a frame)
        monkeypatch.setenv("G1S_W_WGS_C", "24" if reuse == "1" else "264")
        g = DiffGenerator(Fraction(24, 1), 10, 10, batch_frames=3)
        for s, d in frames:
            g.diff_frame(Frame(s, 1, 1), Frame(d, 1, 1))
        g.sync()
        r = g.last_record()
        sums = [r.ar_sums(c) for c in range(3)]
        out[reuse] = (sums, format_tbl(g.finish()))
    for c in range(3):
        a, b = out["1"][0][c], out["0"][0][c]
        assert a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), f"plane {c}"
    assert out["1"][1] == out["0"][1]
    # and against the oracle
    from tests.oracle_binding import OracleDiff, format_tbl as ofmt

    o = OracleDiff(24, 1, 10, 10, 3, True)
    for s, d in frames:
        o.diff_frame(s, d, 1, 1)
    assert out["1"][1] == ofmt(o.finish())


def test_native_shard_driver_over_rccl():
    """tools/shard_native.cpp: the round protocol of include/g1s_diff.h driven from C++ -- one process, one generator per
    visible device, the per-round gather over RCCL (ncclSend / ncclRecv in a group, ncclCommInitAll) -- gives the table of a
    single generator byte for byte (a scene change inside the video, a last batch that is short)."""
    import os
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "shard_native")
    if not os.path.exists(exe):
        pytest.skip("tools/shard_native not built (make -C grav1synth_amd/csrc)")
    r = subprocess.run([exe, "0", "38", "4"], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "IDENTICAL" in r.stdout


@pytest.mark.parametrize("seed", [3, 77])
def test_ragged_sizes_with_isolated_large_residuals(seed):
    """Random frame sizes whose last block column is partial (a window that ends at the plane's right edge leaves
    whole 8-sample words of a multiplied block outside it: they must reach the matrix cores as zeros, not as what
    the LDS held before), all lags and chroma formats, a few |src - den| > 127 samples on the denoised side:
    every integer sum of every frame's record equals the oracle's."""
    import random

    from tests.oracle_binding import OracleDiff

    rng = random.Random(seed)
    for k in range(30):
        w, h = rng.randint(66, 300), rng.randint(66, 300)
        bd = rng.choice([8, 10])
        xd, yd = rng.choice([(1, 1), (1, 0), (0, 0)])
        lag = rng.choice([3, 2, 1])
        spec = SynthSpec(w, h, bd, xdec=xd, ydec=yd, textured=rng.random() < 0.6)
        s, d = np_pair(spec, k)
        d = [p.copy() for p in d]
        nr = np.random.default_rng(rng.randint(0, 1 << 30))
        for c in range(3):
            hh, ww = d[c].shape
            for _ in range(int(nr.integers(0, 4))):
                y, x = int(nr.integers(0, hh)), int(nr.integers(0, ww))
                d[c][y, x] = 0 if (int(s[c][y, x]) >> (bd - 8)) > 140 else (255 << (bd - 8))
        o = OracleDiff(24, 1, bd, bd, lag, True)
        try:
            o.diff_frame(s, d, xd, yd)
        except RuntimeError:
            continue  # (no flat block at all: the oracle refuses like the reference)
        g = DiffGenerator(Fraction(24, 1), bd, bd, ar_coeff_lag=lag, batch_frames=1)
        try:
            g.diff_frame(Frame(s, xd, yd), Frame(d, xd, yd))
            g.sync()
            r = g.last_record()
            for c in range(3):
                S, Sb, n = o.ar_sums(c)
                S2, Sb2, n2 = r.ar_sums(c)
                assert n == n2 and np.array_equal(S, S2) and np.array_equal(Sb, Sb2), f"case {k}: {w}x{h} {bd}b xd{xd} yd{yd} lag{lag}, plane {c}"
        finally:
            g.close()


@pytest.mark.parametrize("lag", [3, 2])
def test_partial_last_column_with_many_units_per_workgroup(lag, monkeypatch):
    """The same property where a workgroup of the accumulation pass walks many units (few workgroups per frame):
    a plain unit before a unit at the plane's right edge leaves its residuals in the LDS words the edge unit masks."""
    from tests.oracle_binding import OracleDiff, format_tbl as ofmt

    monkeypatch.setenv("G1S_F_WGS", "8")
    spec = SynthSpec(1000, 360, 10, xdec=1, ydec=1, textured=False)  # 31.25 blocks wide: windows 8 - lag wide in the last column
    o = OracleDiff(24, 1, 10, 10, lag, True)
    g = DiffGenerator(Fraction(24, 1), 10, 10, ar_coeff_lag=lag, batch_frames=2)
    for k in range(2):
        s, d = np_pair(spec, k)
        o.diff_frame(s, d, 1, 1)
        g.diff_frame(Frame(s, 1, 1), Frame(d, 1, 1))
    g.sync()
    r = g.last_record()
    for c in range(3):
        S, Sb, n = o.ar_sums(c)
        S2, Sb2, n2 = r.ar_sums(c)
        assert n == n2 and np.array_equal(S, S2) and np.array_equal(Sb, Sb2), f"plane {c}"
    assert format_tbl(g.finish()) == ofmt(o.finish())


def test_int32_accumulators_at_their_ceiling(monkeypatch):
    """VERDICT r02 #6a: kMMaxUnits = 240 is sized so that a wave's int32 accumulator cannot overflow (240 units x 512
    samples x 127^2 < 2^31).  Here every workgroup of the accumulation pass walks 220 units (3520 x 1024, 8 workgroups a
    frame) of a frame whose residual is +-127 on EVERY sample, in a checkerboard -- so the products of a matrix entry all
    have one sign and the sums go where the bound says they may: |S| up to 220 x 512 x 127^2 = 1.8e9 per wave.  Every block
    has the same content, hence the same score: the 90th-percentile rule marks them all.  The AR system of such a frame is
    singular (the fold refuses it): the generator only keeps the records, which must hold the oracle's sums."""
    from tests.oracle_binding import OracleDiff

    monkeypatch.setenv("G1S_F_WGS", "8")
    W, H = 3520, 1024
    yy, xx = np.mgrid[0:H, 0:W]
    sign = (1 - 2 * ((xx + yy) & 1)).astype(np.int16)
    den_y = np.full((H, W), 128, np.uint8)
    src_y = (128 + 127 * sign).astype(np.uint8)
    den_c = np.full((H // 2, W // 2), 128, np.uint8)
    src_c = (128 + 127 * sign[: H // 2, : W // 2]).astype(np.uint8)
    src, den = [src_y, src_c, src_c.copy()], [den_y, den_c, den_c.copy()]
    o = OracleDiff(24, 1, 8, 8, 3, True)
    try:
        o.diff_frame(src, den, 1, 1)
    except RuntimeError:
        pass  # (the singular AR system; the sums below were taken before the solve)
    assert (o.flat_mask() != 0).all()
    g = DiffGenerator(Fraction(24, 1), 8, 8, batch_frames=1, records_only=True)
    g.diff_frame(Frame(src, 1, 1), Frame(den, 1, 1))
    recs, n = g.take_records(W, H, 3, 1)
    assert n == 1
    from grav1synth_amd.diff import Record

    r = Record(recs[0])
    assert np.array_equal(r.flat_mask(), o.flat_mask())
    S, Sb, nobs = o.ar_sums(0)  # (the oracle stops at the luma solve: its chroma sums of this frame do not exist)
    S2, Sb2, nobs2 = r.ar_sums(0)
    assert nobs == nobs2 and nobs > 0
    assert np.array_equal(S, S2) and np.array_equal(Sb, Sb2)
    assert np.abs(S).max() == nobs * 127 * 127  # (the diagonal: every product + 127^2)
    g.close()
    # the chroma planes at +-127 on every sample under an ordinary luma plane (the chroma solve falls back, no error)
    spec = SynthSpec(W, H, 8, textured=False)
    s0, d0 = np_pair(spec, 0)
    src, den = [s0[0], src_c, src_c.copy()], [d0[0], den_c, den_c.copy()]
    o = OracleDiff(24, 1, 8, 8, 3, True)
    o.diff_frame(src, den, 1, 1)
    g = DiffGenerator(Fraction(24, 1), 8, 8, batch_frames=1, records_only=True)
    g.diff_frame(Frame(src, 1, 1), Frame(den, 1, 1))
    recs, n = g.take_records(W, H, 3, 1)
    r = Record(recs[0])
    for c in range(3):
        S, Sb, nobs = o.ar_sums(c)
        S2, Sb2, nobs2 = r.ar_sums(c)
        assert nobs == nobs2 and nobs > 0, f"plane {c}"
        assert np.array_equal(S, S2) and np.array_equal(Sb, Sb2), f"plane {c}"
    assert np.abs(r.ar_sums(1)[0]).max() >= (r.ar_sums(1)[2] - 1) * 127 * 127 // 2
    g.close()


@pytest.mark.parametrize("spec,nframes", [(SynthSpec(320, 200, 10), 5), (SynthSpec(288, 160, 8, xdec=0, ydec=0), 4)],
                         ids=["10b420", "8b444"])
def test_y4m_files_give_the_table_of_the_in_memory_path_and_of_the_oracle(tmp_path, spec, nframes):
    """`grav1synth diff SOURCE DENOISED -o OUT` (src/main.rs:414-531) through g1s_diff_y4m_files: pinned
    read-ahead reader, frame-pair loop, finish, table -- same bytes as feeding the frames by hand and as
    the oracle."""
    from grav1synth_amd.ingest import diff_y4m_files, write_y4m

    pairs = [make_pair(spec, k, device="cpu") for k in range(nframes)]
    fps = Fraction(30000, 1001)
    write_y4m(str(tmp_path / "src.y4m"), [s for s, _ in pairs], spec.bit_depth, spec.xdec, spec.ydec, fps)
    write_y4m(str(tmp_path / "den.y4m"), [d for _, d in pairs], spec.bit_depth, spec.xdec, spec.ydec, fps)
    out = tmp_path / "out.tbl"
    frames, unequal = diff_y4m_files(str(tmp_path / "src.y4m"), str(tmp_path / "den.y4m"), str(out), batch_frames=2)
    assert (frames, unequal) == (nframes, False)
    g = DiffGenerator(fps, spec.bit_depth, spec.bit_depth)
    for s, d in pairs:
        g.diff_frame(s, d, spec.xdec, spec.ydec)
    by_hand = format_tbl(g.finish())
    g.close()
    assert out.read_bytes() == by_hand
    oracle_tbl, _ = oracle_run(spec, list(range(nframes)), 3, True, fps=fps)
    assert by_hand == oracle_tbl


def test_diff_command_takes_two_decoders_output_from_pipes(tmp_path):
    """No libav here: the stand-in for the reference's two `BitstreamReader`s is two decoders piping YUV4MPEG2 in
    (`ffmpeg -i source.mkv -f yuv4mpegpipe - > fifo_a`, likewise the denoised file).  The whole command over two FIFOs fed by
    two writer threads, the streams ending at different times as decoders do: table == the files' table."""
    import os
    import threading

    from grav1synth_amd.ingest import diff_y4m_files, write_y4m

    spec = SynthSpec(640, 360, 10)
    nframes = 9
    pairs = [make_pair(spec, k, device="cpu") for k in range(nframes)]
    fps = Fraction(24, 1)
    files = {"src": tmp_path / "src.y4m", "den": tmp_path / "den.y4m"}
    write_y4m(str(files["src"]), [s for s, _ in pairs], spec.bit_depth, spec.xdec, spec.ydec, fps)
    write_y4m(str(files["den"]), [d for _, d in pairs], spec.bit_depth, spec.xdec, spec.ydec, fps)
    want = tmp_path / "files.tbl"
    assert diff_y4m_files(str(files["src"]), str(files["den"]), str(want), batch_frames=4) == (nframes, False)
    fifos = {k: tmp_path / f"{k}.pipe" for k in files}
    for f in fifos.values():
        os.mkfifo(f)

    def feed(name, chunk):
        with open(files[name], "rb") as src, open(fifos[name], "wb") as dst:
            while True:
                b = src.read(chunk)
                if not b:
                    break
                dst.write(b)

    ts = [threading.Thread(target=feed, args=("src", 1 << 16), daemon=True), threading.Thread(target=feed, args=("den", 12345), daemon=True)]
    for t in ts:
        t.start()
    got = tmp_path / "pipes.tbl"
    assert diff_y4m_files(str(fifos["src"]), str(fifos["den"]), str(got), batch_frames=4) == (nframes, False)
    for t in ts:
        t.join(timeout=20)
        assert not t.is_alive()
    assert got.read_bytes() == want.read_bytes() and len(want.read_bytes()) > 100


@pytest.mark.parametrize("devices,nframes", [([0], 7), ([0, 0], 7), ([0, 0], 9), ([0, 0, 0], 11), ("visible", 13)],
                         ids=["1gen", "2gen_7", "2gen_9", "3gen_11", "all_visible_devices"])
def test_sharded_y4m_diff_gives_the_table_of_one_generator(tmp_path, devices, nframes):
    """`diff --gpus N SOURCE DENOISED -o OUT` (g1s_diff_y4m_files_sharded; the loop of src/main.rs:414-531 dealt over
    N generators in one process, batches of 2 -> short last batch, idle generators in the last round, a scene cut inside a
    batch): the same bytes as one generator and as the oracle -- with one generator per visible device (N = 1 on the
    test box) and with several generators on device 0."""
    from grav1synth_amd import cli
    from grav1synth_amd.ingest import diff_y4m_files, write_y4m

    if devices == "visible":
        devices = list(range(torch.cuda.device_count()))
    a, b = SynthSpec(320, 200, 10), SynthSpec(320, 200, 10, gain_scale=3)
    specs = [a] * (nframes // 2) + [b] * (nframes - nframes // 2)
    pairs = [make_pair(sp, k, device="cpu") for k, sp in enumerate(specs)]
    fps = Fraction(30000, 1001)
    write_y4m(str(tmp_path / "src.y4m"), [s for s, _ in pairs], 10, 1, 1, fps)
    write_y4m(str(tmp_path / "den.y4m"), [d for _, d in pairs], 10, 1, 1, fps)
    out = tmp_path / "out.tbl"
    frames, unequal = diff_y4m_files(str(tmp_path / "src.y4m"), str(tmp_path / "den.y4m"), str(out), batch_frames=2, devices=devices)
    assert (frames, unequal) == (nframes, False)
    want, segs = oracle_run(a, range(nframes), fps=fps, specs_per_frame=specs)
    assert len(segs) >= 2
    assert out.read_bytes() == want
    # the front door: python -m grav1synth_amd diff ... --devices 0,0
    out2 = tmp_path / "out2.tbl"
    assert cli.main(["diff", str(tmp_path / "src.y4m"), str(tmp_path / "den.y4m"), "-o", str(out2), "--devices",
                     ",".join(str(d) for d in devices)]) == 0
    assert out2.read_bytes() == want


def test_y4m_unequal_frame_counts_stop_at_the_shorter_file(tmp_path):
    """The reference warns and stops when only one reader ends (src/main.rs:449-455): the table then covers
    the common prefix."""
    from grav1synth_amd.ingest import diff_y4m_files, write_y4m

    spec = SynthSpec(320, 192, 8)
    pairs = [make_pair(spec, k, device="cpu") for k in range(4)]
    write_y4m(str(tmp_path / "src.y4m"), [s for s, _ in pairs], 8, 1, 1)
    write_y4m(str(tmp_path / "den.y4m"), [d for _, d in pairs[:3]], 8, 1, 1)
    out = tmp_path / "out.tbl"
    frames, unequal = diff_y4m_files(str(tmp_path / "src.y4m"), str(tmp_path / "den.y4m"), str(out))
    assert (frames, unequal) == (3, True)
    g = DiffGenerator(Fraction(24, 1), 8, 8)
    for s, d in pairs[:3]:
        g.diff_frame(s, d, 1, 1)
    assert out.read_bytes() == format_tbl(g.finish())
    g.close()
    # geometry mismatch between the two files: diff_frame's error ends the command
    other = SynthSpec(352, 192, 8)
    write_y4m(str(tmp_path / "den2.y4m"), [make_pair(other, 0, device="cpu")[1]], 8, 1, 1)
    with pytest.raises(RuntimeError, match="dimensions do not match"):
        diff_y4m_files(str(tmp_path / "src.y4m"), str(tmp_path / "den2.y4m"), str(out))


def test_streaming_shards_over_rccl_with_one_rank():
    """The frame-shard exchange on the real backend ("nccl" = RCCL), world size 1 (two ranks on one device are
    refused by RCCL; the 2-rank logic is covered over gloo in tests/test_dist_cpu.py): device-side message,
    rooted gather, merge thread -- the table must equal the plain generator's."""
    import os
    import subprocess
    import sys

    code = (
        "import os, torch, torch.distributed as dist\n"
        "from fractions import Fraction\n"
        "from grav1synth_amd.diff import DiffGenerator, format_tbl\n"
        "from grav1synth_amd.dist import StreamingShardedDiff\n"
        "from grav1synth_amd.synth import SynthSpec, make_pair\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "spec = SynthSpec(352, 208, 10)\n"
        "pairs = [make_pair(spec, k, device='cuda') for k in range(11)]\n"
        "sd = StreamingShardedDiff(Fraction(24, 1), 10, 10, device=0, batch_frames=2, group=dist)\n"
        "for i in range(0, 11, 2):\n"
        "    sd.diff_prepared(DiffGenerator.prepare_frames(pairs[i:i + 2], 1, 1))\n"
        "a = format_tbl(sd.finish()); sd.close()\n"
        "g = DiffGenerator(Fraction(24, 1), 10, 10)\n"
        "for s, d in pairs: g.diff_frame(s, d, 1, 1)\n"
        "b = format_tbl(g.finish()); g.close()\n"
        "dist.destroy_process_group()\n"
        "assert a == b and len(a) > 100\n"
        "print('RCCL_OK')\n"
    )
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


# ---- the per-frame half of the fold on the device (csrc/latest.hip, G1S_LATEST=device) ------------------------------------
_LATEST_CASES = [
    (SynthSpec(320, 192, 8), 3, True),
    (SynthSpec(326, 198, 8), 3, True),                       # partial edge blocks in both directions
    (SynthSpec(352, 208, 10), 3, True),
    (SynthSpec(320, 192, 10, xdec=1, ydec=0), 3, True),      # 4:2:2
    (SynthSpec(256, 160, 10, xdec=0, ydec=0), 2, True),      # 4:4:4, lag 2
    (SynthSpec(320, 192, 8), 2, False),                      # luma only
    (SynthSpec(300, 180, 8, textured=False), 1, True),       # all flat, lag 1
    (SynthSpec(1920, 1080, 8), 3, True),
    (SynthSpec(3840, 2160, 10), 3, True),                    # the bench workload
]


def _latest_blobs(monkeypatch, where, spec, lag, chroma, frames, batch):
    monkeypatch.setenv("G1S_LATEST", where)
    g = DiffGenerator(Fraction(24, 1), spec.bit_depth, spec.bit_depth, ar_coeff_lag=lag, luma_only=not chroma,
                      batch_frames=batch, records_only=2)
    for k in frames:
        s, d = make_pair(spec, k, device="cuda")
        g.diff_frame(s, d, spec.xdec, spec.ydec)
    blobs = g.take_latest(len(frames) + 8, sync=True).copy()
    g.close()
    return blobs


@pytest.mark.parametrize("case", _LATEST_CASES, ids=lambda c: f"{c[0].width}x{c[0].height}_{c[0].bit_depth}b_{c[0].xdec}{c[0].ydec}_lag{c[1]}_{'yuv' if c[2] else 'y'}")
def test_device_latest_states_are_the_host_halfs_bytes(monkeypatch, case):
    """k4_latest (AR solve, block measurements, strength solve of every frame on the device) must write the blob the host
    half makes from the same record -- every f64 bit of every system, solution, gain and total, and the header."""
    spec, lag, chroma = case
    frames = list(range(5)) if spec.width < 3000 else [0, 1, 2]
    host = _latest_blobs(monkeypatch, "host", spec, lag, chroma, frames, 3)
    dev = _latest_blobs(monkeypatch, "device", spec, lag, chroma, frames, 3)
    assert host.shape == dev.shape and host.shape[0] == len(frames)
    for i in range(len(frames)):
        if not np.array_equal(host[i], dev[i]):
            bad = np.flatnonzero(host[i] != dev[i])
            raise AssertionError(f"frame {i}: {bad.size} bytes differ, first at {bad[0]} (of {host.shape[1]})")


def test_device_latest_reports_the_host_halfs_failures(monkeypatch):
    """Frames the per-frame half refuses -- one block only ("Not enough flat blocks ..."), a constant frame (singular luma
    system) -- give the same blob (status, message, the state as far as it got) and the same error from a folding generator."""
    from grav1synth_amd._lib import G1SError

    one_block = [np.full((32, 32), 9, np.uint8), np.full((16, 16), 9, np.uint8), np.full((16, 16), 9, np.uint8)]
    constant = [np.full((64, 64), 7, np.uint8), np.full((32, 32), 7, np.uint8), np.full((32, 32), 7, np.uint8)]
    for planes, code in ((one_block, -3), (constant, -4)):
        blobs = {}
        for where in ("host", "device"):
            monkeypatch.setenv("G1S_LATEST", where)
            g = DiffGenerator(Fraction(24, 1), 8, 8, batch_frames=1, records_only=2)
            g.diff_frame(planes, planes)
            blobs[where] = g.take_latest(4, sync=True).copy()
            g.close()
            g = DiffGenerator(Fraction(24, 1), 8, 8, batch_frames=1)
            with pytest.raises(G1SError) as e:
                g.diff_frame(planes, planes)
                g.sync()
            assert e.value.code == code
            blobs[where + "_msg"] = e.value.message
            g.close()
        assert np.array_equal(blobs["host"], blobs["device"])
        assert blobs["host_msg"] == blobs["device_msg"]
        assert int(np.frombuffer(blobs["host"][0].tobytes()[12:16], np.int32)[0]) == code


@pytest.mark.parametrize("spec,lag", [(SynthSpec(320, 192, 8), 3), (SynthSpec(352, 208, 10), 2), (SynthSpec(1920, 1080, 10), 3)])
def test_device_latest_gives_the_oracles_table(monkeypatch, spec, lag):
    """The whole front door with the per-frame half on the device: table == oracle, a scene cut inside a batch included,
    and the statistics the host half used to count."""
    monkeypatch.setenv("G1S_LATEST", "device")
    ks = [0, 1, 2, 50, 51, 3, 4] if spec.width < 1000 else [0, 1, 2]
    tbl, _ = oracle_run(spec, ks, lag)
    g = DiffGenerator(Fraction(24, 1), spec.bit_depth, spec.bit_depth, ar_coeff_lag=lag, batch_frames=3)
    g.set_timing(True)
    for k in ks:
        s, d = make_pair(spec, k, device="cuda")
        g.diff_frame(s, d, spec.xdec, spec.ydec)
    out = format_tbl(g.finish())
    st = g.stats()
    kt = g.kernel_times()
    g.close()
    assert out == tbl
    assert st.frames == len(ks) and st.flat_blocks > 0
    assert "k4_latest" in kt and kt["k4_latest"][1] == (len(ks) + 2) // 3, kt  # (the kernel did run: one launch a batch)
    print({k: round(v[0] / v[1] * 1e3, 1) for k, v in kt.items()})
