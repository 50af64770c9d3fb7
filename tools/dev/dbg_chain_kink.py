import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch as th
from sbmc_amd import functions as funcs
from test_gpu_pointwise_chain import _make, _chain64
dev = th.device("cuda")
(b, s, cin, couts, hw, t_mode, acts, mean) = (4, 2, 128, (128, 128, 128), 204, 2, (1, 1, 0), True)
x, t, layers = _make(b, s, cin, couts, hw, t_mode, acts, 1.0, dev, 77)
ref = _chain64(x, t, s, layers)
ys, signs, amaxes, ymean = funcs.pointwise_chain_forward(x, t, s, layers, store_mid=True, want_signs=True, mean=mean)
for l in range(2):
    mism = ((ys[l] > 0) != (ref[l] > 0))
    print("layer", l, "mask mismatches", int(mism.sum()), "values there", ref[l][mism].tolist(), ys[l][mism].tolist())
