"""tools/bench_fold.py -- host-side rate of the ordered merge (g1s_fold_push_latest) on repeated latest states."""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractions import Fraction
from grav1synth_amd.synth import SynthSpec
from grav1synth_amd.diff import RecordFold, latest_from_records
from tests.helpers import oracle_run, record_from_oracle
spec=SynthSpec(320,192,8)
recs=[]
def collect(o,k): recs.append(record_from_oracle(o,spec,3,3).buf.copy())
oracle_run(spec,[0,1,2,3],3,True,collect=collect)
R=np.stack(recs)
blobs=latest_from_records(R,3)
print(blobs.shape)
big=np.concatenate([blobs]*2500)  # 10000 frames
f=RecordFold(Fraction(24,1),3)
t0=time.perf_counter(); f.push_latest_many(big); dt=time.perf_counter()-t0
print("push_latest us/frame", dt/len(big)*1e6)
f.finish(); f.close()
