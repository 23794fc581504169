import sys, numpy as np
sys.path.insert(0,'.')
from tscd_amd import _host as H
from scipy.optimize import linear_sum_assignment
z=np.load('tests/golden/cafm_cost_device_sample.npz')
crop=z['cost_f16_bits'].view(np.float16).astype(np.float64)
rng=np.random.default_rng(7)
mats=[];cols=[]
for k in range(8):
    n=500
    ri=rng.integers(0,crop.shape[0],n); cj=rng.integers(0,crop.shape[1],n)
    c=crop[np.ix_(ri,cj)]+0.01*rng.standard_normal((n,n))
    m=np.ascontiguousarray(c.astype(np.float16).view(np.uint16))
    r,cc=linear_sum_assignment(m.view(np.float16).astype(np.float64))
    mats.append(m); cols.append(cc.astype(np.int64))
np.stack(mats).tofile('tools/lap_ab/mats.bin'); np.stack(cols).tofile('tools/lap_ab/cols.bin')
print("ok")
