#!/usr/bin/env python3
"""Phase breakdown of the forward / backward kernels from in-kernel s_memtime stamps.
Needs the diagnostic library:  IQLHIP_LIB=jsrl-corl_amd/libiqlhip_stamps.so python tools/gpu_stamps.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import contextlib
import io

import numpy as np
import torch

import iql
import synth

S, A, B = int(os.environ.get("S", 17)), int(os.environ.get("A", 6)), int(os.environ.get("B", 256))
N = 1_000_000
buf = iql.ReplayBuffer(S, A, N, "cuda")
with contextlib.redirect_stdout(io.StringIO()):
    buf.load_d4rl_dataset(synth.synth_transitions(N, S, A, seed=0))
qf, vf, actor = iql.TwinQ(S, A).cuda(), iql.ValueFunction(S).cuda(), iql.GaussianPolicy(S, A, 1.0).cuda()
tr = iql.ImplicitQLearning(1.0, actor, torch.optim.Adam(actor.parameters(), lr=3e-4), qf,
                           torch.optim.Adam(qf.parameters(), lr=3e-4), vf, torch.optim.Adam(vf.parameters(), lr=3e-4),
                           max_steps=1000000, device="cuda")
if os.environ.get("PRECISION"):
    tr.train_on_buffer(buf, B, seed=7)
    tr.set_precision(os.environ["PRECISION"])
for it in range(int(os.environ.get("ITERS", 50))):        # (ITERS=3000: the chip's clocks have ramped by then)
    tr.train_on_buffer(buf, B, seed=7)
torch.cuda.synchronize()
raw = tr.debug_read("stamps")
st = raw.view(np.uint64).reshape(4096, 16)


def report(name, blocks, labels, sel=None):
    blk = st[blocks]
    blk = blk[blk[:, 0] > 0]
    if sel is not None:
        blk = blk[sel(blk)]
    if len(blk) == 0:
        print(name, "no blocks")
        return
    t0 = blk[:, 0].astype(np.int64)
    first = t0.min()
    print(f"{name}: {len(blk)} blocks; start skew (cycles) median {np.median(t0 - first):.0f} max {np.max(t0 - first)}")
    prev = 0
    for i, lab in labels:
        d = blk[:, i].astype(np.int64) - blk[:, prev].astype(np.int64)
        ok = blk[:, i] > 0
        if ok.sum() == 0:
            continue
        print(f"   {lab:28s} median {np.median(d[ok]):8.0f}  p90 {np.percentile(d[ok], 90):8.0f}  max {d[ok].max():8d} cycles")
        prev = i
    last = max(i for i, _ in labels)
    tot = blk[:, last].astype(np.int64) - t0
    print(f"   {'block total':28s} median {np.median(tot):8.0f}  max {tot.max():8d};  kernel span {int((blk[:, last].astype(np.int64)).max() - first)} cycles")


def realtime_report(name, blocks):
    """Entry (slot 14) and exit (slot 15) on the chip-wide 100 MHz clock: when blocks start and end relative to the
    first block of the launch (10 ns ticks -> us)."""
    blk = st[blocks]
    blk = blk[(blk[:, 14] > 0) & (blk[:, 15] > 0)]
    if len(blk) == 0:
        print(name, "no realtime stamps")
        return
    t0 = blk[:, 14].astype(np.int64)
    t1 = blk[:, 15].astype(np.int64)
    z = t0.min()
    q = lambda a, p_: np.percentile(a, p_) / 100.0
    print(f"{name}: {len(blk)} blocks;  start after first block: median {q(t0 - z, 50):.2f} p90 {q(t0 - z, 90):.2f} max {q(t0 - z, 100):.2f} us;"
          f"  duration: median {q(t1 - t0, 50):.2f} max {q(t1 - t0, 100):.2f} us;  end after first start: median {q(t1 - z, 50):.2f} "
          f"p90 {q(t1 - z, 90):.2f} max {q(t1 - z, 100):.2f} us")


def xcd_spans(name, blocks, last):
    """s_memtime counters are per XCD: start skew and span are only meaningful inside one XCD (block id % 8)."""
    out = []
    for x in range(8):
        blk = st[blocks[(blocks & 7) == x]]
        blk = blk[blk[:, 0] > 0]
        if len(blk) == 0:
            continue
        t0 = blk[:, 0].astype(np.int64)
        t1 = blk[:, last].astype(np.int64)
        t1 = np.where(t1 > 0, t1, t0)
        out.append((x, len(blk), int(np.median(t0 - t0.min())), int((t0 - t0.min()).max()), int((t1 - t0).max()), int(t1.max() - t0.min())))
    print(f"{name}: per XCD (blocks, start skew median/max, longest block, span first-start..last-end)")
    for o in out:
        print("   xcd %d: %3d blocks  skew %6d / %6d   longest %6d   span %6d" % o)


n_rt = (B + 31) // 32
fsl2 = int(os.environ.get("IQLHIP_FWD_SPB_L2", 0 if 8 * n_rt * 4 <= 256 else (1 if 8 * n_rt * 2 <= 256 else 2)))
fwd_blocks = np.arange(8 * 2 * ((n_rt + 1) // 2) if fsl2 == 2 else 8 * n_rt * (4 >> fsl2))
# iql_fwd_kernel's block map: XCD pair (x & 3) hosts two instances, bit 0 of the block's index on its XCD says which
FWD_NAMES = ("V(s)", "Q1", "Q2", "PI", "V(s')", "Qt1", "Qt2", "idle")
fwd_inst = np.where(((fwd_blocks >> 3) & 1) == 0, fwd_blocks & 3, 4 + (fwd_blocks & 3))
report("fwd slice 0 (after the layer-0 barrier)", fwd_blocks, [(2, "(layer-0 barrier)"), (8, "H0 save (+ next slice's W2/b1 requests)"), (9, "layer-1 operand reads + MFMAs"), (10, "epilogue + H1 tile write"), (11, "barrier"),
                                                              (12, "H1 save + head partials"), (13, "barrier before the next slice")])
report("fwd", fwd_blocks, [(1, "prefetch+gather"), (5, "L0: LDS operand reads"), (6, "L0: MFMAs"), (7, "L0: epilogue + H0 writes"), (2, "L0: barrier"), (3, "H0 save + layer1"), (4, "H1 save + head")])
n_chunk = (B + 255) // 256
# launch_bwd's layout: beyond two rounds of the chip the (b) blocks walk 4 column slices and take the FIRST block indices
bsl2 = int(os.environ.get("IQLHIP_BWD_SPB_L2", 2 if 4 * (32 * n_chunk + 4 * n_rt) > 512 else 0))
n_b = (4 >> bsl2) * n_rt
per_net = 32 * n_chunk + n_b
nb = 8 * ((per_net + 1) // 2)
bw = 2048 + np.arange(nb)
ids = np.arange(nb)
local = (ids >> 3) * 2 + ((ids & 7) >> 2)
net_of = ids & 3
if bsl2 > 0:         # MULTI instantiation: (b) blocks first (donated policy tiles, if any, are not tracked here)
    local = np.where(local < n_b, 32 * n_chunk + local, local - n_b)
a_blocks = bw[local < 32 * n_chunk]
b_blocks = bw[(local >= 32 * n_chunk) & (local < per_net)]
LAB_A = [(10, "issue loads"), (5, "pi: row loads + w"), (6, "pi: wS barrier"), (7, "pi: (row,dim) math"), (11, "wait heads + dY (rest)"), (1, "LDS zero/barrier"), (2, "designated b2/log_std"), (3, "loads + 128 MFMA"),
         (4, "reduce + store (+extras)")]
LAB_B = [(5, "W1 prefetch+gather+dY(32)"), (6, "dH1 tile"), (7, "128 MFMA + red write"), (8, "reduce+mask"), (12, "dW0: LDS reads + MFMA"), (13, "dW0: stage T + barrier"), (9, "dW0: copy out")]
if os.environ.get("PRECISION") == "bf16" and bsl2 > 0:      # the full-width bf16 tail of the (b) blocks
    LAB_B = [(5, "loads + dY(32)"), (6, "dH1 tile + barrier"), (7, "dH0: 64 MFMAs over all k"), (8, "mask + transposed tile + barrier"), (12, "dW0 operands (X side)"), (9, "dW0 MFMAs + stores")]
if os.environ.get("PER_IT"):
    for n, nm in ((0, "V"), (3, "PI")):
        for itv in range(4):
            report(f"bwd (a) net {nm} it=={itv}", bw[(local < 32 * n_chunk) & (net_of == n) & ((local & 3) == itv)], LAB_A)
if os.environ.get("PER_NET"):
    for n, nm in enumerate(("V", "Q1", "Q2", "PI")):
        isn = net_of == n
        report(f"bwd (a) net {nm}", bw[(local < 32 * n_chunk) & isn], LAB_A)
        report(f"bwd (b) net {nm}", bw[(local >= 32 * n_chunk) & (local < per_net) & isn], LAB_B)
    for i, nm in enumerate(FWD_NAMES[:7]):
        report(f"fwd inst {nm}", fwd_blocks[fwd_inst == i], [(1, "prefetch+gather"), (2, "layer0"), (3, "H0 save + layer1"), (4, "H1 save + head")])
report("bwd (a) dW1 tiles", a_blocks, [(10, "issue loads"), (11, "wait heads + dY"), (1, "LDS zero/barrier"), (2, "designated b2/log_std"), (3, "loads + 128 MFMA"),
                                        (4, "reduce + store (+extras)")])
report("bwd (b) dH0/dW0", b_blocks, [(5, "W1 prefetch+gather+dY(32)"), (6, "dH1 tile"), (7, "128 MFMA + red write"),
                                      (8, "reduce+mask"), (12, "dW0: LDS reads + MFMA"), (13, "dW0: stage T + barrier"), (9, "dW0: copy out")])
realtime_report("fwd  all", fwd_blocks[fwd_inst != 7])
realtime_report("bwd  (a)", a_blocks)
realtime_report("bwd  (b)", b_blocks)
realtime_report("bwd  all", bw)
for n, nm in enumerate(("V", "Q1", "Q2", "PI")):
    realtime_report(f"bwd (b) {nm}", bw[(local >= 32 * n_chunk) & (local < per_net) & (net_of == n)])
    realtime_report(f"bwd (a) {nm}", bw[(local < 32 * n_chunk) & (net_of == n)])
    sel = (local < 32 * n_chunk) & (net_of == n)
    for itv in range(4):
        realtime_report(f"bwd (a) {nm} it=={itv}", bw[sel & ((local & 3) == itv)])
for i, nm in enumerate(FWD_NAMES[:7]):
    realtime_report(f"fwd inst {nm}", fwd_blocks[fwd_inst == i])
# clock estimate: cycles per 100 MHz tick over the fwd kernel
blk = st[fwd_blocks]
blk = blk[blk[:, 0] > 0]
print("stamp[15] (realtime) range", int(blk[:, 15].max() - blk[:, 15].min()), "ticks of 10ns vs cycles", int(blk[:, 0].max() - blk[:, 0].min()))
