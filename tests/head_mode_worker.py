"""Child process of tests/test_hip_dp.py::test_head_modes_and_forward_layouts_agree: one short train_steps run under whatever
IQLHIP_* switches the parent set (they are read when the library is loaded); dumps the losses and a parameter checksum.

    python head_mode_worker.py <out.json>
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import numpy as np
    import iql
    import synth
    from hip_helpers import build_hip_trainer, read_params
    S, A, N, B = 17, 6, 5000, 256
    params = synth.synth_params(S, A, seed=5)
    data = synth.synth_transitions(N, S, A, seed=6)
    buf = iql.ReplayBuffer(S, A, N, "cuda")
    buf.load_d4rl_dataset({k: v.copy() for k, v in data.items()})
    tr = build_hip_trainer(params, S, A, True, {"iql_tau": 0.7, "beta": 3.0, "discount": 0.99, "tau": 0.005},
                           {"v": 3e-4, "q": 3e-4, "pi": 3e-4}, 1000)
    losses = [tr.train_steps(buf, n, B, seed=9) for n in (1, 20, 7, 70)]      # head of 1, 4 and 2 steps; odd and even calls
    pr = read_params(tr)
    out = {"losses": np.concatenate(losses).astype(np.float64).tolist(),
           "params": {f"{n}.{k}": float(np.asarray(v, dtype=np.float64).sum()) for n, t in pr.items() for k, v in t.items()}}
    json.dump(out, open(sys.argv[1], "w"))


if __name__ == "__main__":
    main()
