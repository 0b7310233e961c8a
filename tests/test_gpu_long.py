"""BASELINE.json configs[3] at its stated size -- diff 3840x2160 10-bit 4:2:0, lag 3, chroma, 1000 frames sharded eight ways -- and configs[4]'s
format in eight shards -- 7680x4320 10-bit 4:4:4, 32 frames, four a shard, cuts at frame 16 (a shard boundary) and 22 (inside one) -- on the
ONE device a test box has, against a table the oracle wrote for the same seeded frames (tests/golden/make_golden.py long: an hour
and a half of oracle time, committed as data).  Two scene cuts: frame 500 (a boundary of the 125-frame shards, inside a 64-frame
batch) and frame 768 (a boundary of the 64-frame batches, inside a 125-frame shard).  The reference's loop is strictly ordered
(/root/reference/src/main.rs:432-521): whatever deals the frames, the table must be the ordered job's, byte for byte.

(Both jobs run all of what follows; sizes below are the 4K job's.)

  * one generator, 1000 frames in order                                          (the plain path at this length)
  * eight gloo ranks sharing device 0: the streaming job with the per-frame half on the host and on the device, and contiguous
    125-frame shards with one exchange at the end                               (tests/dist_gpu_long_worker.py)
  * the command itself: two YUV4MPEG2 streams of 1000 frames through FIFOs into g1s_diff_y4m_files_sharded with eight generators
    on device 0 (`python -m grav1synth_amd diff ... --devices 0,0,0,0,0,0,0,0`)
"""
import os
import queue
import socket
import subprocess
import sys
import threading

import pytest
import torch

from grav1synth_amd.diff import DiffGenerator, format_tbl
from grav1synth_amd.synth import make_pair
from tests.golden import make_golden

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["oracle_full_3840x2160_10b_420_lag3_1000frames.tbl", "oracle_full_7680x4320_10b_444_lag3_32frames.tbl"]


@pytest.fixture(scope="module", params=NAMES, ids=["4K_420_1000_frames", "8K_444_32_frames"])
def job(request):
    """(golden's name, its job, the frames' specs, fps, the golden's bytes)"""
    from fractions import Fraction

    name = request.param
    g = make_golden.LONG[name]
    with open(os.path.join(ROOT, "tests", "golden", name), "rb") as f:
        tbl = f.read()
    assert tbl.count(b"\nE ") == 3, "the golden holds three segments (two scene cuts)"
    return name, g, make_golden.frame_specs(g), Fraction(*g["fps"]), tbl


def _keep(name, tbl):
    """G1S_LONG_OUT=dir: the tables of these jobs are also left there (to look at a mismatch off the box)"""
    d = os.environ.get("G1S_LONG_OUT")
    if d:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name), "wb") as f:
            f.write(tbl)


def test_one_generator_over_the_whole_job_matches_the_oracle_golden(job):
    name, g, specs, fps, want = job
    spec = g["spec"]
    gen = DiffGenerator(fps, spec.bit_depth, spec.bit_depth, ar_coeff_lag=g["lag"])  # (the engine's own launch group: 64 frames at 4K)
    keep = []
    for k, sp in enumerate(specs):
        s, d = make_pair(sp, k, device="cuda")
        gen.diff_frame(s, d, spec.xdec, spec.ydec)
        keep.append((s, d))  # (device frames are read in place: alive until the generator has released them)
        if len(keep) > 512:
            gen.sync()
            del keep[:256]
    got = format_tbl(gen.finish())
    gen.close()
    del keep
    torch.cuda.empty_cache()
    _keep("one_generator.tbl", got)
    assert got == want


def test_eight_ranks_on_one_device_match_the_oracle_golden(tmp_path, job):
    name, g, specs, fps, want = job
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(8):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="8", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   G1S_FOLD_THREADS="2", OMP_NUM_THREADS="1", GPU_MAX_HW_QUEUES="8", G1S_LONG_NAME=name)
        env.pop("G1S_LATEST", None)
        procs.append(subprocess.Popen([sys.executable, "-m", "tests.dist_gpu_long_worker", str(tmp_path)], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(x[-1500:] for x in logs)
    for name in ("streaming_host.tbl", "streaming_device.tbl", "contiguous.tbl"):
        _keep(name, (tmp_path / name).read_bytes())
    for name in ("streaming_host.tbl", "streaming_device.tbl", "contiguous.tbl"):
        assert (tmp_path / name).read_bytes() == want, name


def _y4m_frame(planes) -> bytes:
    return b"FRAME\n" + b"".join(p.cpu().numpy().astype("<u2", copy=False).tobytes() for p in planes)


def test_the_sharded_command_over_two_pipes_matches_the_oracle_golden(tmp_path, job):
    from grav1synth_amd.ingest import diff_y4m_files

    name, g, specs, fps, want = job
    spec = g["spec"]
    fifos = {k: str(tmp_path / f"{k}.pipe") for k in ("src", "den")}
    for f in fifos.values():
        os.mkfifo(f)
    qs = {k: queue.Queue(maxsize=6) for k in fifos}
    failed = []

    def produce():
        try:
            for k, sp in enumerate(specs):
                s, d = make_pair(sp, k, device="cuda")
                qs["src"].put(_y4m_frame(s))
                qs["den"].put(_y4m_frame(d))
        except Exception as e:  # (the readers see a short stream and the command fails with its own message)
            failed.append(e)
        finally:
            for q in qs.values():
                q.put(None)

    def feed(name):
        import fcntl

        with open(fifos[name], "wb", buffering=0) as dst:
            try:
                fcntl.fcntl(dst.fileno(), 1031, 1 << 20)  # F_SETPIPE_SZ
            except OSError:
                pass
            dst.write(f"YUV4MPEG2 W{spec.width} H{spec.height} F{fps.numerator}:{fps.denominator} Ip A1:1 C{'420' if spec.xdec else '444'}p10\n".encode())
            while True:
                b = qs[name].get()
                if b is None:
                    break
                mv = memoryview(b)
                while len(mv):
                    mv = mv[dst.write(mv):]

    ts = [threading.Thread(target=produce, daemon=True)] + [threading.Thread(target=feed, args=(k,), daemon=True) for k in fifos]
    for t in ts:
        t.start()
    out = tmp_path / "out.tbl"
    frames, unequal = diff_y4m_files(fifos["src"], fifos["den"], str(out), devices=[0] * 8, batch_frames=int(g.get("batch", 0)))
    for t in ts:
        t.join(timeout=60)
        assert not t.is_alive()
    assert not failed, failed
    assert (frames, unequal) == (g["frames"], False)
    _keep("command_8_generators.tbl", out.read_bytes())
    assert out.read_bytes() == want
