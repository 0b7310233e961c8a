#!/usr/bin/env python3
"""tools/est_pmc.py -- two 32-frame launches of the estimate kernel over one 4K 10-bit plane (for rocprofv3 --pmc runs: few other kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grav1synth_amd.estimate import NoiseEstimator
p = (torch.rand((2160, 3840), device="cuda") * 40 + 300).to(torch.int16).view(torch.uint16) if hasattr(torch, "uint16") else None
torch.cuda.synchronize()
est = NoiseEstimator(10, batch_frames=32)
for k in range(64):
    est.estimate_frame(p)
print(est.finish()[:2])
est.close()
