"""tools/footprint.py -- device memory one generator holds (six slots of the engine's own launch group), by format and by where the per-frame half runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fractions import Fraction
from grav1synth_amd.diff import DiffGenerator
from grav1synth_amd.synth import SynthSpec, make_pair
for (w,h,bd,x,y,nm) in ((3840,2160,10,1,1,"4K 10-bit 4:2:0"),(1920,1080,8,1,1,"1080p 8-bit 4:2:0"),(7680,4320,10,0,0,"8K 10-bit 4:4:4")):
    for where in ("host","device"):
        os.environ["G1S_LATEST"]=where
        spec=SynthSpec(w,h,bd,x,y)
        s,d=make_pair(spec,0,device="cuda"); torch.cuda.synchronize()
        f0,_=torch.cuda.mem_get_info()
        g=DiffGenerator(Fraction(24,1),bd,bd)
        for k in range(2): g.diff_frame(s,d,x,y)
        g.sync(); f1,_=torch.cuda.mem_get_info()
        print(nm, "per-frame half on the", where, ": device bytes held by one generator", round((f0-f1)/2**20), "MiB", flush=True)
        g.finish(); g.close()
