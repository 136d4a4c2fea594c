#!/usr/bin/env python3
"""GPU timeline of the driver's 20-step call.  Run under `rocprofv3 --kernel-trace --output-format csv`; prints the host
clock (MONOTONIC / BOOTTIME / REALTIME, ns) around each timed call so that the kernel trace can be laid beside it
(tools/gpu_call_timeline.py --report <kernel_trace.csv> <this script's stdout> does that)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--report":
    import csv
    rows = []
    with open(sys.argv[2]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
    rows.sort()
    calls = []
    for line in open(sys.argv[3]):
        if line.startswith("CALL"):
            p = line.split()
            calls.append({"n": int(p[1]), "t0": [int(x) for x in p[2:5]], "ret": [int(x) for x in p[5:8]], "t1": [int(x) for x in p[8:11]]})
    # which host clock is the trace's: the one that puts the call's kernels between t0 and t1
    dom = None
    for d in range(3):
        c = calls[-1]
        inside = [r for r in rows if c["t0"][d] <= r[0] and r[1] <= c["t1"][d]]
        if len(inside) >= 3 * c["n"]:
            dom = d
            break
    print("trace clock domain:", ["MONOTONIC", "BOOTTIME", "REALTIME"][dom] if dom is not None else "none matched")
    if dom is None:
        sys.exit(1)
    for c in calls[-3:]:
        ks = [r for r in rows if c["t0"][dom] <= r[0] and r[1] <= c["t1"][dom]]
        t0 = c["t0"][dom]
        print(f"--- call of {c['n']} steps: host returned at {(c['ret'][dom]-t0)/1e3:.1f} us, synchronize returned at {(c['t1'][dom]-t0)/1e3:.1f} us; {len(ks)} kernels")
        prev_end = t0
        for i, (s, e, name) in enumerate(ks):
            tag = ""
            if i < 14 or i >= len(ks) - 4 or (s - prev_end) > 1500:
                print(f"  {i:3d} {name:40s} start {(s-t0)/1e3:7.1f}  dur {(e-s)/1e3:5.2f}  gap before {(s-prev_end)/1e3:5.2f}")
            prev_end = e
        print(f"  last kernel ends at {(ks[-1][1]-t0)/1e3:.1f} us; first starts at {(ks[0][0]-t0)/1e3:.1f} us; "
              f"sum of durations {sum(e-s for s,e,_ in ks)/1e3:.1f} us; sum of gaps {(ks[-1][1]-ks[0][0]-sum(e-s for s,e,_ in ks))/1e3:.1f} us")
    sys.exit(0)
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import __graft_entry__ as ge
ge.build()
import iql, synth
from hip_helpers import build_hip_trainer
S, A, B, N = 17, 6, 256, 1_000_000
data = synth.synth_transitions(N, S, A, seed=0)
buf = iql.ReplayBuffer(S, A, N, "cuda")
buf.load_d4rl_dataset(data)
params = synth.synth_params(S, A, seed=1)
tr = build_hip_trainer(params, S, A, True, {"iql_tau": .7, "beta": 3., "discount": .99, "tau": .005}, {"v": 3e-4, "q": 3e-4, "pi": 3e-4}, 1_000_000)
tr.prepare_train_steps(buf, B)
tr.train_steps(buf, 5, B, seed=1234, return_losses=False); torch.cuda.synchronize()
def clocks():
    return (time.clock_gettime_ns(time.CLOCK_MONOTONIC), time.clock_gettime_ns(time.CLOCK_BOOTTIME), time.clock_gettime_ns(time.CLOCK_REALTIME))
for n in (20, 20, 20, 20, 20, 20):
    torch.cuda.synchronize()
    a = clocks()
    tr.train_steps(buf, n, B, seed=1234, return_losses=False)
    b = clocks()
    torch.cuda.synchronize()
    c = clocks()
    print("CALL", n, *a, *b, *c, f"# {(c[0]-a[0])/1e3:.1f} us", flush=True)
