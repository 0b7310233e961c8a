"""CPU suite, part 2: the product's host logic without a GPU -- the C-ABI
library loads and exports every declared symbol, the ordered fold equals the
oracle, the .tbl writer/parser agree with the reference's format sample."""
import ctypes as C
import os
import re
from fractions import Fraction

import numpy as np
import pytest

from grav1synth_amd import _lib
from grav1synth_amd.diff import GrainTableSegment, Record, RecordFold, format_tbl
from grav1synth_amd.synth import SynthSpec
from grav1synth_amd.tbl import TblError, parse_tbl
from tests.helpers import oracle_run, record_from_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "g1s_diff.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(g1s_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    L = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/g1s_diff.h but not exported"
    bound = {s[0] for s in _lib.SYMBOLS}
    assert declared == bound, f"binding/header mismatch: {declared ^ bound}"
    _lib.lib()


def test_no_gpu_means_loud_failure_not_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from grav1synth_amd.diff import DiffGenerator

    with pytest.raises(_lib.G1SError) as e:
        DiffGenerator(Fraction(24, 1), 8, 8)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_product_does_not_reference_the_oracle():
    """The shipped package must never import, link or call anything under oracle/."""
    banned = ("liborc_diff", "oracle_binding", "diff_oracle", "orc_diff", "orc_format", "OracleDiff")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "grav1synth_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                for b in banned:
                    assert b not in txt, f"{f} references {b}"


CASES = [
    (SynthSpec(320, 192, 8), 3, True, 3),
    (SynthSpec(320, 200, 10, xdec=1, ydec=0), 3, True, 2),
    (SynthSpec(256, 160, 8, xdec=0, ydec=0), 2, True, 2),
    (SynthSpec(256, 160, 8), 2, False, 2),
    (SynthSpec(256, 160, 12), 1, True, 2),
    (SynthSpec(216, 152, 10), 3, True, 2),  # cut last row and column: block sample counts that are no powers of two (768, 576, 192, 144)
    # a cut last column / row that luma measures and chroma does not (<= 32 samples at chroma resolution): Cb's list of measured
    # blocks differs from luma's, so Cb builds its own strength matrix instead of copying luma's (fold.cpp compute_latest, like = -1)
    (SynthSpec(228, 152, 10), 3, True, 2),
    (SynthSpec(216, 132, 8), 3, True, 2),
    (SynthSpec(226, 160, 10, xdec=1, ydec=0), 3, True, 2),
    (SynthSpec(228, 132, 8, textured=False), 2, True, 2),
]


@pytest.mark.parametrize("spec,lag,chroma,nframes", CASES)
def test_fold_over_exact_integer_records_equals_oracle(spec, lag, chroma, nframes):
    """The product converts exact integer sums to f64 once, the reference sums
    rounded f64 terms: the emitted table must still be byte-identical."""
    fold = RecordFold(Fraction(24, 1), lag)
    npl = 3 if chroma else 1
    tbl, _ = oracle_run(spec, range(nframes), lag, chroma,
                        collect=lambda o, k: fold.push(record_from_oracle(o, spec, lag, npl).buf))
    assert format_tbl(fold.finish()) == tbl


def test_fold_segmentation_matches_oracle():
    a = SynthSpec(320, 192, 8)
    b = SynthSpec(320, 192, 8, gain_scale=3)
    specs = [a, a, a, b, b, b]
    fps = Fraction(30000, 1001)
    fold = RecordFold(fps, 3)
    tbl, segs = oracle_run(a, range(6), fps=fps, specs_per_frame=specs,
                           collect=lambda o, k: fold.push(record_from_oracle(o, a, 3, 3).buf))
    out = fold.finish()
    assert len(out) == len(segs) >= 2
    assert format_tbl(out) == tbl


def test_finish_with_a_small_buffer_loses_nothing():
    """g1s_fold_finish (same contract as g1s_diff_finish): a buffer that is too small reports the segment count with
    G1S_ERR_CAPACITY and keeps the segments; the second call, sized from the count, gets all of them."""
    from grav1synth_amd._lib import G1SSegment

    a = SynthSpec(320, 192, 8)
    b = SynthSpec(320, 192, 8, gain_scale=3)
    specs = [a, a, a, b, b, b]
    fps = Fraction(30000, 1001)
    fold = RecordFold(fps, 3)
    tbl, segs = oracle_run(a, range(6), fps=fps, specs_per_frame=specs,
                           collect=lambda o, k: fold.push(record_from_oracle(o, a, 3, 3).buf))
    L = _lib.lib()
    n = C.c_size_t()
    one = (G1SSegment * 1)()
    assert L.g1s_fold_finish(fold._h, one, 1, C.byref(n)) == _lib.G1S_ERR_CAPACITY
    assert n.value == len(segs) >= 2
    assert L.g1s_fold_finish(fold._h, None, 0, C.byref(n)) == _lib.G1S_ERR_CAPACITY and n.value == len(segs)
    out = fold.finish()  # the mirror's own retry path works on the same object
    assert format_tbl(out) == tbl
    assert format_tbl(fold.finish()) == tbl  # and finish stays repeatable


def test_fold_rejects_bad_records():
    fold = RecordFold(Fraction(24, 1), 3)
    with pytest.raises(_lib.G1SError):
        fold.push(np.zeros(16, np.uint8))
    r = Record.blank(64, 64, 1, 1, 3, 2)  # lag mismatch
    with pytest.raises(_lib.G1SError):
        fold.push(r.buf)
    r = Record.blank(64, 64, 1, 1, 3, 3)  # no flat blocks
    with pytest.raises(_lib.G1SError) as e:
        fold.push(r.buf)
    assert e.value.code == -3 and "Not enough flat blocks" in e.value.message


def test_tbl_writer_reproduces_the_reference_sample_byte_for_byte():
    """tests/golden/reference-example-table.tbl is the reference's own data file
    (tests/example-table.tbl): parse -> write must give the same bytes, incl. the
    double space after `sY 14` (src/main.rs:659) and empty `cY` (lag 0)."""
    raw = open(os.path.join(ROOT, "tests", "golden", "reference-example-table.tbl"), "rb").read()
    segs = parse_tbl(raw)
    assert len(segs) == 1 and segs[0].ar_coeff_lag == 0 and segs[0].random_seed == 7391
    assert len(segs[0].scaling_points_y) == 14 and segs[0].ar_coeffs_cb == [0]
    assert format_tbl(segs) == raw


def test_tbl_round_trip_of_diff_output_and_parser_errors():
    tbl, _ = oracle_run(SynthSpec(192, 128, 8), range(1))
    assert format_tbl(parse_tbl(tbl)) == tbl
    with pytest.raises(TblError):
        parse_tbl(b"filmgrn2\n")
    with pytest.raises(TblError):
        parse_tbl(tbl.replace(b"\tcY", b"\tcY 1"))  # coefficient count no longer 2*lag*(lag+1)


def test_batched_merge_equals_frame_by_frame_merge_across_segment_cuts():
    """g1s_fold_push_latest merges a batch with the combined-model solves taken out of the serial chain
    (speculative prefix sums, discarded behind a segment cut): same table as one frame at a time, as the
    record path (sequential push) and as the oracle, with cuts inside a batch."""
    from grav1synth_amd.diff import latest_from_records

    a = SynthSpec(320, 192, 8)
    b = SynthSpec(320, 192, 8, gain_scale=3)
    specs = [a, a, a, b, b, b, b, a, a, b, a, a]
    fps = Fraction(30000, 1001)
    recs = []
    tbl, segs = oracle_run(a, range(len(specs)), fps=fps, specs_per_frame=specs,
                           collect=lambda o, k: recs.append(record_from_oracle(o, a, 3, 3).buf.copy()))
    assert len(segs) >= 3
    recs = np.stack(recs)
    blobs = latest_from_records(recs, 3)
    tables = []
    for mode in ("all_at_once", "one_by_one", "records", "uneven"):
        fold = RecordFold(fps, 3)
        if mode == "all_at_once":
            fold.push_latest_many(blobs)
        elif mode == "one_by_one":
            for k in range(len(blobs)):
                fold.push_latest_many(blobs[k:k + 1])
        elif mode == "records":
            fold.push_many(recs)
        else:
            fold.push_latest_many(blobs[:5])
            fold.push_latest_many(blobs[5:6])
            fold.push_latest_many(blobs[6:])
        tables.append(format_tbl(fold.finish()))
        fold.close()
    assert all(t == tbl for t in tables)
    # the batch call (g1s_latest_from_records: the process pool) gives the bytes of the one-record call, whatever the strides
    L = _lib.lib()
    bs = int(L.g1s_latest_size(3))
    wide_recs = np.zeros((len(recs), recs.shape[1] + 64), np.uint8)
    wide_recs[:, :recs.shape[1]] = recs
    wide_blobs = np.zeros((len(recs), bs + 24), np.uint8)
    assert L.g1s_latest_from_records(wide_recs.ctypes.data, wide_recs.shape[1], len(recs), 3, wide_blobs.ctypes.data, wide_blobs.shape[1]) == 0
    one = np.zeros(bs, np.uint8)
    for k in range(len(recs)):
        assert L.g1s_latest_from_record(recs[k].ctypes.data, recs.shape[1], 3, one.ctypes.data, bs) == 0
        assert np.array_equal(one, wide_blobs[k, :bs]) and np.array_equal(one, blobs[k])
    assert L.g1s_latest_from_records(wide_recs.ctypes.data, wide_recs.shape[1], 2, 3, wide_blobs.ctypes.data, bs - 8) != 0  # blobs too small
    assert 1 <= int(L.g1s_usable_cpus()) <= (os.cpu_count() or 1)


def test_native_tbl_reader_round_trips_and_matches_the_python_reader():
    """N1: the consumer side of diff's output.  g1s_parse_tbl must read what g1s_format_tbl and the reference
    write (tests/golden/*.tbl incl. the reference's own sample table), agree with the Python reader, and
    reject what it rejects; g1s_tbl_segment_for is `apply`'s per-frame lookup (src/parser/frame.rs:617-633)."""
    import glob
    import os

    from grav1synth_amd.tbl import GrainTable, TblError, parse_tbl, parse_tbl_native

    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.tbl")))
    assert len(files) >= 5
    for f in files:
        data = open(f, "rb").read()
        a, b = parse_tbl(data), parse_tbl_native(data)
        assert a == b and len(a) >= 1
        assert format_tbl(b) == data          # writer(reader(x)) == x, byte for byte
    for bad in (b"", b"filmgrn2\n", b"filmgrn1\nX 0 1 1 1 1\n", b"filmgrn1\nE 0 1 1 1\n", b"filmgrn1\nE 0 1 0 1 1\n",
                b"filmgrn1\nE 0 9 1 7 1\n\tp 3 7 0 11 0 1 128 192 256 128 192 256\n"):
        with pytest.raises(TblError):
            parse_tbl_native(bad)
        with pytest.raises((TblError, ValueError)):
            parse_tbl(bad)
    data = open(os.path.join(os.path.dirname(__file__), "golden", "oracle_scenecut_30000_1001.tbl"), "rb").read()
    segs = parse_tbl_native(data)
    assert len(segs) >= 2
    t = GrainTable(segs)
    first = t.segment_for(0)
    assert first.start_time == 0 and first.random_seed == (segs[0].random_seed + 10956) & 0xffff
    again = t.segment_for(segs[0].end_time - 1)
    assert again.start_time == 0 and again.random_seed == (segs[0].random_seed + 2 * 10956) & 0xffff
    second = t.segment_for(segs[0].end_time)
    assert second.start_time == segs[1].start_time and second.random_seed == (segs[1].random_seed + 10956) & 0xffff
    assert GrainTable(segs[:1]).segment_for(segs[0].end_time) is None


_SHARE_CODE = r"""
import hashlib, sys
import numpy as np
from fractions import Fraction
from grav1synth_amd.diff import latest_from_records
from grav1synth_amd.synth import SynthSpec
from tests.helpers import oracle_run, record_from_oracle
h = hashlib.sha256()
for spec, cut in ((SynthSpec(228, 152, 10), "col"), (SynthSpec(216, 132, 8), "row"), (SynthSpec(226, 160, 10, xdec=1, ydec=0), "col"),
                  (SynthSpec(216, 152, 10), "none")):
    recs = []
    oracle_run(spec, range(2), 3, True, collect=lambda o, k: recs.append(record_from_oracle(o, spec, 3, 3)))
    for r in recs:
        nbw, nbh = r.nbw, r.nbh
        m = r.views(0)["mask"].reshape(nbh, nbw)
        # the cut last column / row: flagged flat, with statistics of its own (4 luma columns or rows: 128 luma samples,
        # 32 -- or 16 -- chroma samples: luma measures the block, the chroma planes do not)
        idx = [] if cut == "none" else [by * nbw + nbw - 1 for by in range(nbh - 1)] if cut == "col" else [(nbh - 1) * nbw + bx for bx in range(nbw - 1)]
        for c in range(3):
            v = r.views(c)
            for j, b in enumerate(idx):
                if c == 0:
                    v["mask"][b] = 1
                    v["luma_sum"][b] = 128 * (90 + 7 * j)
                v["sum_d"][b] = 5 - 3 * j + c
                v["sum_d2"][b] = 400 + 37 * j + 11 * c
        assert cut == "none" or m[:, -1].any() or m[-1, :].any()
    blobs = latest_from_records(np.stack([r.buf for r in recs]), 3)
    h.update(blobs.tobytes())
print(h.hexdigest())
"""


def test_planes_sharing_the_strength_matrix_equals_every_plane_on_its_own():
    """compute_latest's plane sharing (fold.cpp: Cb copies luma's strength matrix when it measures the very blocks luma measures,
    Cr Cb's list and matrix) against every plane building its own (G1S_FOLD_NO_SHARE=1), blob for blob -- on records whose cut
    last column / row luma measures and chroma does not (<= 32 chroma samples: Cb's list differs from luma's, the branch no
    synthetic frame reaches: the finder never marks such a block flat on this content, so the mask is set by hand)."""
    import subprocess
    import sys

    outs = []
    for extra in ({}, {"G1S_FOLD_NO_SHARE": "1"}):
        env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), **extra)
        p = subprocess.run([sys.executable, "-c", _SHARE_CODE], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(p.stdout.strip().splitlines()[-1])
    assert len(outs[0]) == 64 and outs[0] == outs[1]


def test_ctypes_mirrors_have_the_headers_struct_sizes(tmp_path):
    """The Python side fills and reads the ABI's structs through ctypes mirrors (`_lib.py`); `g1s_diff_get_stats` and its
    siblings write sizeof(struct) bytes into them.  A field added to the header and not to the mirror (or the other way
    round) is a memory overrun no test of values would see: compile the header with the C compiler and compare the sizes
    and the offset of each struct's last field."""
    import ctypes as C
    import shutil
    import subprocess

    from grav1synth_amd import _lib

    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    pairs = [("g1s_frame_t", _lib.G1SFrame, "on_device"), ("g1s_segment_t", _lib.G1SSegment, None), ("g1s_opts_t", _lib.G1SOpts, None),
             ("g1s_stats_t", _lib.G1SStats, "chain_batches"), ("g1s_filter_desc_t", _lib.G1SFilterDesc, "alg"),
             ("g1s_y4m_info_t", _lib.G1SY4MInfo, "fps_den")]
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "g1s_diff.h"', "int main(void) {"]
    for name, _, last in pairs:
        src.append(f'  printf("{name} %zu %zu\\n", sizeof({name}), {f"offsetof({name}, {last})" if last else "(size_t)0"});')
    src += ["  return 0;", "}"]
    c_file = tmp_path / "sizes.c"
    c_file.write_text("\n".join(src))
    exe = tmp_path / "sizes"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([cc, "-std=c11", "-I", os.path.join(root, "include"), str(c_file), "-o", str(exe)])
    got = {ln.split()[0]: (int(ln.split()[1]), int(ln.split()[2])) for ln in subprocess.check_output([str(exe)], text=True).splitlines()}
    for name, mirror, last in pairs:
        assert C.sizeof(mirror) == got[name][0], f"{name}: header {got[name][0]} bytes, ctypes mirror {C.sizeof(mirror)}"
        if last:
            assert getattr(mirror, last).offset == got[name][1], f"{name}.{last}: header offset {got[name][1]}, mirror {getattr(mirror, last).offset}"
