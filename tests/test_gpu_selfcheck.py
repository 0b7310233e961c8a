"""GPU self-checks: the HIP path against ITSELF (another chain, another launch geometry, a tuning switch) -- not parity
evidence (tests/test_gpu_parity.py holds every comparison with the oracle), but what keeps the fallback chain and the
switches of DESIGN.md honest.  Everything here runs in its own interpreter: the switches are read once per process."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wide_and_stream_chains_agree():
    """G1S_K3 = wide (default: k3w.hip.h) and stream (round 3's kernel, what the wide chain falls back on for unaligned
    planes, odd widths and mixed depths) give the same records and tables, bit for bit -- small odd formats, 12-bit
    residuals outside int8, the 4K workload; the wide chain also with the fewest and with many workgroups a frame."""
    lines = {}
    for mode, extra in (("wide", {}), ("stream", {"G1S_K3": "stream"}), ("wide-few", {"G1S_W_WGS": "8", "G1S_W_WGS_C": "8"}),
                        ("wide-many", {"G1S_W_WGS": "16384", "G1S_W_WGS_C": "16384"}), ("stream-noreuse", {"G1S_K3": "stream", "G1S_F_REUSE": "0"})):
        env = dict(os.environ, **extra)
        p = subprocess.run([sys.executable, "-m", "tests.k3_mode_digest"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert p.returncode == 0, f"{mode}: {p.stderr[-2000:]}"
        lines[mode] = json.loads(p.stdout.strip().splitlines()[-1])
    for mode in lines:
        assert lines[mode] == lines["wide"], f"wide vs {mode}"


_SWITCH_CODE = (
    "import hashlib, sys\n"
    "from fractions import Fraction\n"
    "from grav1synth_amd.diff import DiffGenerator, format_tbl\n"
    "from grav1synth_amd.synth import SynthSpec, make_pair\n"
    "spec = SynthSpec(352, 208, 10)\n"
    "h = hashlib.sha256()\n"
    "g = DiffGenerator(Fraction(24, 1), 10, 10, batch_frames=2, records_only=True)\n"
    "for k in range(5):\n"
    "    s, d = make_pair(spec, k, device='cuda'); g.diff_frame(s, d, 1, 1)\n"
    "recs, n = g.take_records(spec.width, spec.height, 3, 5); g.close(); h.update(recs.tobytes())\n"
    "g = DiffGenerator(Fraction(24, 1), 10, 10, batch_frames=2)\n"
    "for k in range(5):\n"
    "    s, d = make_pair(spec, k, device='cuda'); g.diff_frame(s, d, 1, 1)\n"
    "h.update(format_tbl(g.finish())); g.close(); print(h.hexdigest())\n"
)


def _run(code, extra):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), **extra)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout.strip().splitlines()[-1]


def test_tuning_switches_do_not_change_results():
    """The environment switches of DESIGN.md (stream layout, launch sizes and frame order of the wide launches, the prefetch
    touches, finder modes, fold threads): records and table must equal the default run's."""
    ref = _run(_SWITCH_CODE, {})
    assert len(ref) == 64
    for extra in ({"G1S_ONE_STREAM": "1"}, {"G1S_NO_DEFER": "1"}, {"G1S_K1_LITERAL": "1"}, {"G1S_K1_LITERAL": "2"}, {"G1S_FOLD_THREADS": "1"},
                  {"G1S_W_REV": "3"}, {"G1S_W_WGS": "4096", "G1S_W_WGS_C": "8"}, {"G1S_W_OFF": "1"},
                  {"G1S_F_SERIAL": "1"}, {"G1S_W_ASIDE": "1"}, {"G1S_SIDE2": "1"}):
        assert _run(_SWITCH_CODE, extra) == ref, extra


def test_mixed_depths_agree_between_the_chains():
    """A 10-bit source against an 8-bit denoised video (and the other way round, and 10 against 12 bits): the wide chain's general
    residual form and the stream chain give the same records and tables (the oracle comparison:
    tests/test_gpu_parity.py::test_mixed_depths_records_match_oracle)."""
    code = (
        "import hashlib\n"
        "from fractions import Fraction\n"
        "from grav1synth_amd.diff import DiffGenerator, format_tbl\n"
        "from grav1synth_amd.synth import SynthSpec, make_pair\n"
        "h = hashlib.sha256()\n"
        "for sb, db, lo in ((10, 8, False), (8, 10, False), (10, 12, False), (10, 8, True)):\n"
        "    ss, sd = SynthSpec(416, 232, sb), SynthSpec(416, 232, db)\n"
        "    pairs = [(make_pair(ss, k, device='cuda')[0], make_pair(sd, k, device='cuda')[1]) for k in range(5)]\n"
        "    g = DiffGenerator(Fraction(24, 1), sb, db, luma_only=lo, batch_frames=2, records_only=True)\n"
        "    for s, d in pairs: g.diff_frame(s, d, 1, 1)\n"
        "    recs, n = g.take_records(416, 232, 1 if lo else 3, 5); g.close(); h.update(recs.tobytes())\n"
        "    g = DiffGenerator(Fraction(24, 1), sb, db, luma_only=lo, batch_frames=2)\n"
        "    for s, d in pairs: g.diff_frame(s, d, 1, 1)\n"
        "    h.update(format_tbl(g.finish())); g.close()\n"
        "print(h.hexdigest())\n"
    )
    ref = _run(code, {})
    assert len(ref) == 64
    assert _run(code, {"G1S_K3": "stream"}) == ref


def test_per_plane_deferrals_agree_between_the_chains():
    """4:4:4 frames whose Cb plane alone, and whose Cr plane alone, holds residuals outside int8: the chroma launch defers
    a unit per PLANE; the records and the table must be the stream chain's (which is held to the oracle on the same
    mechanism in tests/test_gpu_parity.py)."""
    code = (
        "import hashlib, numpy as np\n"
        "from fractions import Fraction\n"
        "from grav1synth_amd.diff import DiffGenerator, format_tbl\n"
        "from grav1synth_amd.synth import SynthSpec, make_pair\n"
        "h = hashlib.sha256()\n"
        "for bd, w, hh in ((10, 352, 224), (8, 288, 160)):\n"
        "    spec = SynthSpec(w, hh, bd, xdec=0, ydec=0, textured=False)\n"
        "    pairs = []\n"
        "    for k in range(4):\n"
        "        s, d = make_pair(spec, k, device='cpu')\n"
        "        s = [np.array(p) for p in s]; d = [np.array(p) for p in d]\n"
        "        if k in (1, 3): d[1][40:44, 70:75] = 0 if bd == 8 else 3   # Cb far off: |d| > 127 after narrowing\n"
        "        if k in (2, 3): d[2][100:103, 200:204] = (255 if bd == 8 else 1020)\n"
        "        pairs.append((s, d))\n"
        "    g = DiffGenerator(Fraction(24, 1), bd, bd, batch_frames=2, records_only=True)\n"
        "    for s, d in pairs: g.diff_frame(s, d, 0, 0)\n"
        "    recs, n = g.take_records(w, hh, 3, 4); g.close(); h.update(recs.tobytes())\n"
        "    g = DiffGenerator(Fraction(24, 1), bd, bd, batch_frames=2)\n"
        "    for s, d in pairs: g.diff_frame(s, d, 0, 0)\n"
        "    h.update(format_tbl(g.finish())); g.close()\n"
        "print(h.hexdigest())\n"
    )
    ref = _run(code, {})
    assert len(ref) == 64
    assert _run(code, {"G1S_K3": "stream"}) == ref
    assert _run(code, {"G1S_K3": "stream", "G1S_F_REUSE": "0"}) == ref


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the way the driver's SCALE tier would start it): bench.py spawns
    its two ranks under torch.distributed.run, rank 0 prints ONE JSON line.  Two ranks on this one GPU (gloo: RCCL refuses two
    ranks on a device)."""
    env = dict(os.environ, G1S_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # (24 batches a rank and step: the feeding thread runs as far ahead of the device as the generator's slots let it, so the flush
    #  rounds behind the last feed have every slot to empty -- with --cycles 2 a rank fed two batches and four rounds were enough
    #  for a generator of six slots)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--cycles", "24",
                        "--frames", "64", "--no-all-flat", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 1 and out["value"] > 0
    assert out["config"]["backend"] == "gloo" and out["scaling"] == "weak"


def test_bench_eight_ranks_on_one_device_end_in_the_one_rank_jobs_table():
    """The rehearsal of `python bench.py --gpus 8` a one-GPU box allows: eight ranks on this device (gloo), the per-frame half of the
    fold on the device (G1S_LATEST=device: what bench.py picks when a 16-core quota gives a rank two cores), three timed steps.
    finish() refuses a job whose merged frame count is not what the ranks fed; the line's tbl_sha256 (every timed step ended in
    these bytes) must be the one-rank job's over the same video: rank r's 64 resident frames are frames 64 r .. 64 r + 63 of a
    512-frame cycle, so `--gpus 1 --frames 512` feeds the same frames in the same order."""
    def line(gpus, frames, extra_env):
        env = dict(os.environ, **extra_env)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1", "--cycles", "3",
                            "--frames", str(frames), "--no-all-flat", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=1200,
                           cwd=ROOT)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, p.stdout[-2000:]
        return json.loads(lines[0])

    eight = line(8, 64, {"G1S_BENCH_SHARE_GPU": "1", "G1S_LATEST": "device"})
    assert eight["n_gpus"] == 8 and eight["config"]["backend"] == "gloo" and eight["config"]["per_frame_fold_half"].startswith("device")
    assert eight["config"]["frames_per_rank_per_step"] == 192 and eight["value"] > 0
    one = line(1, 512, {})
    assert one["tbl_sha256"] == one["tbl_sha256_untimed_half_batch_job"]
    assert eight["tbl_sha256"] == one["tbl_sha256"] and eight["tbl_bytes"] == one["tbl_bytes"] > 100


@pytest.mark.parametrize("half", ["host", "device"])
def test_bench_one_rank_through_the_rccl_code_path(half):
    """G1S_BENCH_FORCE_DIST=1: ONE rank through everything `bench.py --gpus N` does over RCCL -- the process group on "nccl", the
    streaming frame shards with the rounds' transport (pinned rings, the gather on its own stream), the barrier and the MAX
    all-reduce of the timed region on DEVICE tensors -- which the two-rank test above (two ranks on one GPU: gloo, CPU tensors)
    cannot reach.  (A variable of the N > 1 branch once shadowed the flag that picks CPU tensors for gloo: the all-reduce of
    the step time would have been handed a CPU tensor on RCCL.  Nothing on a one-GPU box ran that line.)  The table of the
    job is the one-process job's: `value` > 0 and a finished line."""
    # (half = "device": RCCL rounds AND k4_latest together -- what every rank of `bench.py --gpus 8` runs on a 16-core quota)
    env = dict(os.environ, G1S_BENCH_FORCE_DIST="1", G1S_LATEST=half)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "G1S_BENCH_SHARE_GPU"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--cycles", "24",
                        "--frames", "64", "--no-all-flat", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0
    assert out["config"]["backend"] == "nccl" and out["config"]["rccl_ranks"] == 1
    assert out["config"]["per_frame_fold_half"].startswith("device" if half == "device" else "host")
    assert out["roofline"]["frac"] > 0
    # (the table every timed step ended in is the one-process job's over the same frames: the digest of the 4K bench content, 64 frames x 24 cycles)
    assert len(out["tbl_sha256"]) == 64


def test_chain_timing_brackets_a_batch_and_changes_nothing():
    """g1s_diff_set_timing(g, 2): one pair of HIP events around a batch's whole chain of kernels (bench.py's roofline.frac).
    Every timed batch is counted, the bracket is no longer than the sum of the per-kernel event pairs of the same batches
    (each of which puts a barrier packet between two launches), and the table is the untimed job's."""
    from fractions import Fraction

    from grav1synth_amd.diff import DiffGenerator, format_tbl
    from grav1synth_amd.synth import SynthSpec, make_pair

    spec = SynthSpec(1280, 704, 10)
    pairs = [make_pair(spec, k, device="cuda") for k in range(8)]
    out = {}
    for mode in (False, True, 2):
        g = DiffGenerator(Fraction(24, 1), 10, 10, batch_frames=4)
        for s, d in pairs[:4]:  # (one untimed batch first: the first launch of a kernel in a process loads its code object)
            g.diff_frame(s, d, 1, 1)
        g.sync()
        g.set_timing(mode)
        for rep in range(3):
            for s, d in pairs:
                g.diff_frame(s, d, 1, 1)
        g.sync()
        st = g.stats()
        kt = g.kernel_times() if mode is True else {}
        out[mode] = (format_tbl(g.finish()), st.ms_chain, st.chain_batches, sum(v[0] for v in kt.values()))
        g.close()
    assert out[False][0] == out[True][0] == out[2][0]
    assert out[False][2] == 0 and out[True][2] == 0, "only set_timing(2) counts chain batches"
    assert out[2][2] == 6 and out[2][1] > 0.0
    assert out[2][1] < out[True][3] * 1.10, (out[2][1], out[True][3])
