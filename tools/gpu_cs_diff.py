import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/jsrl-corl_amd", "/root/repo/tests"]
import numpy as np, torch
import synth
from hip_helpers import build_hip_trainer as build, to_torch_batch as to_tb
S, A, B = 39, 28, 1024
params = synth.synth_params(S, A, seed=1900, gaussian=True)
d = synth.synth_transitions(B, S, A, seed=1901)
batch = {"s": d["observations"], "a": d["actions"], "r": d["rewards"], "ns": d["next_observations"], "d": d["terminals"]}
hyper = {"iql_tau": 0.8, "beta": 3.0, "discount": 0.99, "tau": 0.005, "deterministic": False}
lrs = {"v": 3e-4, "q": 3e-4, "pi": 3e-4}
outs = []
for cs in ("1", "0", "1"):
    os.environ["IQLHIP_LB_CSPLIT"] = cs
    tr = build(params, S, A, True, hyper, lrs, 1000)
    tr.set_precision("bf16")
    outs.append(tr.flat_gradient(to_tb(batch)).copy())
    L = tr._layout
a, b, c = outs
print("cs1 vs cs1 equal:", np.array_equal(a, c))
idx = np.nonzero(a != b)[0]
print("differing entries:", len(idx), idx[:20])
import iqlhip_binding as hb
for i, n in enumerate(("V", "Q1", "Q2", "PI")):
    net = L.net[i]
    print(n, {k: getattr(net, k) for k in ("seg_begin", "w1", "w0", "b0", "b1", "w2", "b2", "log_std")})
print("rel diffs:", np.abs(a[idx] - b[idx])[:10], a[idx][:10])
