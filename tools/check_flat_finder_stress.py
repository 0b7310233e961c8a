#!/usr/bin/env python3
"""tools/check_flat_finder_stress.py [frames] [seed] -- the certified flat-block finder against the literal kernel on
content built to sit ON the decision boundaries: per 32x32 block a random plane (offset, x / y slope) plus noise whose
sigma is drawn log-uniformly across the range where the four thresholds and the 90th-percentile cut decide, plus
blocks of constant colour, saturated blocks, single-pixel spikes and checkerboards.  Sizes with partial right /
bottom blocks, 8 / 10 / 12 bit.  Mask bytes and f32 score bits must agree for every block."""
import sys
from fractions import Fraction

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from grav1synth_amd.diff import DiffGenerator  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
gen = torch.Generator(device="cuda").manual_seed(seed)
bad = tot = lit = 0
for (W, H, bd) in ((3840, 2160, 10), (1937, 1111, 8), (2050, 1170, 12), (1280, 720, 10)):
    nbw, nbh = (W + 31) // 32, (H + 31) // 32
    frames = []
    for k in range(n):
        def blockwise(v):  # [nbh, nbw] -> [H, W]
            return v.repeat_interleave(32, 0).repeat_interleave(32, 1)[:H, :W]
        r = lambda *s: torch.rand(*s, device="cuda", generator=gen)  # noqa: E731
        off = blockwise(r(nbh, nbw) * 200 + 20)
        sx = blockwise((r(nbh, nbw) - 0.5) * 2.0)
        sy = blockwise((r(nbh, nbw) - 0.5) * 2.0)
        sigma = blockwise(torch.exp(r(nbh, nbw) * 5.0 - 2.5))          # 0.08 .. 12 levels (8-bit units)
        kind = blockwise(torch.floor(r(nbh, nbw) * 12))               # 0..11: mostly noise planes, some specials
        yy, xx = torch.meshgrid(torch.arange(H, device="cuda") % 32, torch.arange(W, device="cuda") % 32, indexing="ij")
        img = off + sx * (xx - 16) + sy * (yy - 16) + sigma * torch.randn(H, W, device="cuda", generator=gen)
        img = torch.where(kind == 0, off, img)                                        # constant colour
        img = torch.where(kind == 1, torch.full_like(img, 255.0), img)                # saturated
        img = torch.where((kind == 2) & (xx == 7) & (yy == 9), img + 90, img)         # a spike
        img = torch.where(kind == 3, off + ((xx + yy) % 2) * sigma, img)              # checkerboard
        v8 = img.round().clamp(0, 255)
        # sub-8-bit detail must not matter (the reference truncates with >> (bd - 8)): random low bits
        lo = torch.floor(r(H, W) * (1 << (bd - 8))) if bd > 8 else torch.zeros_like(v8)
        y = (v8 * (1 << (bd - 8)) + lo).to(torch.int32)
        dt = torch.uint8 if bd == 8 else torch.int16
        src = [y.to(dt).contiguous()]
        den = [(y - (torch.randn(H, W, device="cuda", generator=gen) * 2 * (1 << (bd - 8))).round().to(torch.int32)).clamp(0, (1 << bd) - 1).to(dt).contiguous()]
        if bd > 8:
            src = [p.view(torch.uint16) for p in src]
            den = [p.view(torch.uint16) for p in den]
        frames.append((src, den))
    out = []
    for mode in (0, 1, 2):
        g = DiffGenerator(Fraction(24, 1), bd, bd, batch_frames=1, luma_only=True)
        g.set_flat_finder(mode)
        g.set_timing(True)
        rs = []
        for s, d in frames:
            try:
                g.diff_frame(s, d, 1, 1)
                g.sync()
            except Exception:
                pass  # (a fold error of synthetic garbage does not matter here: the record is there)
            r = g.last_record()
            rs.append((r.flat_mask().copy(), r.scores().view(np.uint32).copy()))
        if mode == 0:
            lit += g.stats().literal_blocks
        out.append(rs)
        try:
            g.close()
        except Exception:
            pass
    for a, b, c in zip(*out):
        bad += int((a[0] != b[0]).sum()) + int((a[1] != b[1]).sum()) + int((c[0] != b[0]).sum()) + int((c[1] != b[1]).sum())
        tot += a[0].size
print(f"blocks {tot}  mismatches {bad}  sent to the literal kernels by the fast path {lit} ({100.0 * lit / max(tot, 1):.3f} %)")
sys.exit(1 if bad else 0)
