import os, sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
from smokephysai_amd.models.encoder import HipEncoder
enc = HipEncoder(bench.encoder_weights(0))
x = torch.rand(64, 256, 256, device='cuda') * 1.5
def t(mode, reps=30):
    os.environ['SMK_ENC_PRIO'] = str(mode)
    enc.tokens(x, dtype='bf16x3'); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): enc.tokens(x, dtype='bf16x3')
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for _ in range(20): enc.tokens(x, dtype='bf16x3')
res = {m: [] for m in (0, 1, 2, 3)}
for rnd in range(5):
    for m in (0, 1, 2, 3):
        res[m].append(t(m))
for m, v in res.items(): print('prio mode', m, 'median %.4f min %.4f' % (np.median(v), min(v)), ['%.3f' % a for a in v])
