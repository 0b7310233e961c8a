"""Deterministic synthetic source/denoised frame pairs.

Integer-only arithmetic on int64 torch tensors, so the same call produces the
same bits on CPU and on `cuda:N` (there is no network for real footage, and the
reference has no sample clips for `diff`).  Layout follows what
`BitstreamReader::decode_frame` hands to the estimator (reference
src/reader.rs:183-209): planar Y/U/V, u8 for 8-bit and little-endian u16 for
9..16-bit, chroma decimated by (xdec, ydec).

Content:
  denoised = smooth ramp (+ an 8x8 checker "texture" over the right third of the
             frame, which the flat-block finder must reject)
  noise    = Irwin-Hall(4 hash bytes), spatially correlated by a small symmetric
             integer filter, intensity-dependent gain, chroma = own + luma/4
  source   = clamp(denoised + noise)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch

DEFAULT_SEED = 0x67726176  # "grav"
_M32 = 0xFFFFFFFF


def _fmix32(h: torch.Tensor) -> torch.Tensor:
    """murmur3 32-bit finaliser on non-negative int64 tensors holding u32 values."""
    h = h ^ (h >> 16)
    h = (h * 0x85EBCA6B) & _M32
    h = h ^ (h >> 13)
    h = (h * 0xC2B2AE35) & _M32
    h = h ^ (h >> 16)
    return h


def _irwin_hall(seed: int, frame: int, plane: int, w: int, h: int, device) -> torch.Tensor:
    """n0(x, y) in [-510, 510] on a (h+2, w+2) grid whose [1:-1,1:-1] part is the plane."""
    ys = torch.arange(-1, h + 1, device=device, dtype=torch.int64).view(-1, 1)
    xs = torch.arange(-1, w + 1, device=device, dtype=torch.int64).view(1, -1)
    k = ((ys + 1) * (w + 2) + (xs + 1)) & _M32
    salt = (seed + frame * 0x85EBCA77 + plane * 0xC2B2AE3D) & _M32
    hsh = _fmix32(((k * 0x9E3779B1) + salt) & _M32)
    b = (hsh & 0xFF) + ((hsh >> 8) & 0xFF) + ((hsh >> 16) & 0xFF) + ((hsh >> 24) & 0xFF)
    return b - 510


def _corr_noise(n0: torch.Tensor) -> torch.Tensor:
    """n = 4*n0(x,y) + the four axial neighbours (isotropic under 90-degree
    rotations, so the flat-block finder's eigenvalue-ratio test passes), on the
    plane proper.  sigma(n) = sqrt(20) * 147.8 ~= 661."""
    c = n0[1:-1, 1:-1]
    return 4 * c + n0[1:-1, :-2] + n0[1:-1, 2:] + n0[:-2, 1:-1] + n0[2:, 1:-1]


@dataclass
class SynthSpec:
    width: int
    height: int
    bit_depth: int = 8
    xdec: int = 1
    ydec: int = 1
    textured: bool = True  # False = the all-flat stress variant
    seed: int = DEFAULT_SEED
    gain_scale: int = 1  # scene-cut variant: double the noise gain
    nplanes: int = 3


def plane_dims(spec: SynthSpec, c: int) -> Tuple[int, int]:
    if c == 0:
        return spec.width, spec.height
    return spec.width >> spec.xdec, spec.height >> spec.ydec


def make_pair(spec: SynthSpec, frame: int, device="cpu") -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """Returns (source_planes, denoised_planes); each a list of 2-D tensors
    (uint8 for 8-bit, int16-viewable uint16 stored as torch.int16 is avoided:
    we use torch.uint16 when bit_depth > 8)."""
    W, H, bd = spec.width, spec.height, spec.bit_depth
    up = bd - 8
    maxv = (1 << bd) - 1
    dt = torch.uint8 if bd == 8 else torch.uint16
    src: List[torch.Tensor] = []
    den: List[torch.Tensor] = []

    ys = torch.arange(H, device=device, dtype=torch.int64).view(-1, 1)
    xs = torch.arange(W, device=device, dtype=torch.int64).view(1, -1)
    # luma base in 8-bit units: 32 .. 192 ramp
    base8 = 32 + (xs * 120) // W + (ys * 40) // H
    if spec.textured:
        tex = (xs >= (2 * W) // 3)
        checker = ((((xs >> 3) + (ys >> 3)) & 1) * 48 - 24) + ((xs & 1) * 8)
        base8 = base8 + tex * checker
    base = (base8 << up).expand(H, W)
    n0 = _irwin_hall(spec.seed, frame, 0, W, H, device)
    n = _corr_noise(n0)
    gain = (2 + (base8 >> 6)) * spec.gain_scale  # 2..5
    lnoise = (n * gain) >> (10 - up)
    d_y = base.clamp(0, maxv)
    s_y = (base + lnoise).clamp(0, maxv)
    den.append(d_y.to(dt).contiguous())
    src.append(s_y.to(dt))
    # true luma noise after clamping, for the chroma cross term
    lres = s_y - d_y

    if spec.nplanes == 3:
        cw, ch = plane_dims(spec, 1)
        cys = torch.arange(ch, device=device, dtype=torch.int64).view(-1, 1)
        cxs = torch.arange(cw, device=device, dtype=torch.int64).view(1, -1)
        lco = lres[:: (1 << spec.ydec), :: (1 << spec.xdec)][:ch, :cw]
        for c in (1, 2):
            cb8 = 128 + ((cxs * 24) // cw if c == 1 else -((cys * 24) // ch))
            cbase = (cb8 << up).expand(ch, cw)
            cn0 = _irwin_hall(spec.seed, frame, c, cw, ch, device)
            cn = _corr_noise(cn0)
            cnoise = ((cn * 3 * spec.gain_scale) >> (10 - up)) + (lco >> 2)
            d_c = cbase.clamp(0, maxv)
            s_c = (cbase + cnoise).clamp(0, maxv)
            den.append(d_c.to(dt).contiguous())
            src.append(s_c.to(dt))
    return src, den
