"""GPU parity tests of the LARGE-BATCH bf16 path (jsrl-corl_amd/csrc/iqlhip_lb_kernels.h: more than 512 rows per step in
bf16 precision — BASELINE configs[4]'s per-GPU share of 1 024 rows and its whole 8 192-row batch on one GPU).

Checker: the oracle (oracle/iql_oracle.py, pinned to the reference's own outputs by tests/test_oracle_golden.py) on the
same seeded inputs.  Tolerances are the bf16 statement of SURVEY §8d / north_star: losses rel <= 5e-3 against the fp32
values; gradients within 8.5e-2 in relative L2 norm per tensor (bf16 operands carry 8 significant bits), head-bias /
log_std gradients — means of signed residuals, heavy cancellation — within 2e-2 of the residual scale.  Determinism:
the multi-step driver's chunk graphs equal eager steps on the same indices bit for bit, as on the fp32 path.
"""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def _hip():
    from hip_helpers import build_hip_trainer, read_params, to_torch_batch, unflatten_grads
    return build_hip_trainer, read_params, to_torch_batch, unflatten_grads


def _case(S, A, B, seed, gaussian=True):
    params = synth.synth_params(S, A, seed=seed, gaussian=gaussian)
    d = synth.synth_transitions(B, S, A, seed=seed + 1)
    batch = {"s": d["observations"], "a": d["actions"], "r": d["rewards"], "ns": d["next_observations"],
             "d": d["terminals"]}
    hyper = {"iql_tau": 0.8, "beta": 3.0, "discount": 0.99, "tau": 0.005, "deterministic": not gaussian}
    lrs = {"v": 3e-4, "q": 3e-4, "pi": 3e-4}
    return params, batch, hyper, lrs


# (round 4: tightened from 2e-2 / 1e-1 to 2.5x / 1.13x the worst observed over every bf16 case of the suite: 2.0e-3 / 7.55e-2)
LOSS_RTOL = 5e-3
GRAD_RL2 = 8.5e-2


def _check_grads(got, want_all, tol=GRAD_RL2):
    worst = 0.0
    for net, tensors in want_all.items():
        for t, want in tensors.items():
            g = got[net][t].reshape(want.shape).astype(np.float64)
            if want.size <= 32:
                assert np.max(np.abs(g - want)) <= 2e-2 * max(1.0, float(np.max(np.abs(want)))), (net, t)
                continue
            den = float(np.linalg.norm(want))
            if den < 1e-20:
                continue
            e = float(np.linalg.norm(g - want)) / den
            worst = max(worst, e)
            assert e <= tol, (net, t, e)
    return worst


@pytest.mark.parametrize("S,A,B,gaussian", [
    (39, 28, 1024, True),      # configs[4]'s per-GPU share
    (39, 28, 8192, True),      # configs[4]'s whole batch on one GPU: row blocks walk 8 row tiles, chunk groups of 8 chunks
    (39, 28, 600, True),       # ragged: 19 row tiles (an odd count), a partial chunk, a partial 64-row GEMM stage
    (39, 28, 1000, True),      # ragged last row tile (8 rows)
    (17, 6, 1024, False),      # deterministic policy, one 16-wide head tile, 24-column layer-0 input (one k-block)
    (29, 8, 2080, True),       # antmaze dims, 65 row tiles (more tiles than twice the blocks), 2 k-blocks
])
def test_lb_bf16_step_against_the_oracle(S, A, B, gaussian):
    from oracle import iql_oracle as O
    build, read_params, to_tb, unflat = _hip()
    params, batch, hyper, lrs = _case(S, A, B, seed=500 + B + S, gaussian=gaussian)
    ref = O.iql_losses_and_grads(params, batch, hyper)
    want_l = [ref["value_loss"], ref["q_loss"], ref["actor_loss"]]
    tr = build(params, S, A, gaussian, hyper, lrs, 1000)
    tr.set_precision("bf16")
    tb = to_tb(batch)
    grads, lw = unflat(tr, tr.flat_gradient(tb))
    for got, want in zip(lw, want_l):
        assert abs(got - want) <= LOSS_RTOL * abs(want), (lw, want_l)
    worst = _check_grads(grads, ref["grads"])
    assert worst > 1e-5          # the bf16 path really ran (fp32 would sit at ~1e-7)
    worst_l = max(abs(g - w) / abs(w) for g, w in zip(lw, want_l))
    print(f"S={S} A={A} B={B}: worst relative-L2 gradient error vs oracle {worst:.3e}, worst loss error {worst_l:.3e}")
    log = tr.train(tb)
    for got, want in zip([log["value_loss"], log["q_loss"], log["actor_loss"]], want_l):
        assert abs(got - want) <= LOSS_RTOL * abs(want)
    # the step moved every tensor (the update kernel's large-batch gradient sources are wired to every arena range)
    after = read_params(tr)
    for net in ("vf", "q1", "q2", "pi"):
        for t, before in params[net].items():
            if t == "log_std" and not gaussian:
                continue
            assert np.any(after[net][t].reshape(before.shape) != before), (net, t)


def test_lb_bf16_dropout_and_against_small_batch_kernels(monkeypatch):
    """The large-batch kernels against the small-batch bf16 kernels forced to the same batch (IQLHIP_LB=0), with actor
    dropout by injected masks: two valid bf16 evaluations of one step — losses within 2e-3, parameters within two Adam
    steps' reach after two steps."""
    build, read_params, to_tb, _ = _hip()
    S, A, B = 39, 28, 1024
    params, batch, hyper, lrs = _case(S, A, B, seed=901)
    k0, k1 = synth.synth_dropout_keep(B, 0.1, seed=63)
    outs = []
    for lb in ("1", "0"):
        monkeypatch.setenv("IQLHIP_LB", lb)
        tr = build(params, S, A, True, hyper, lrs, 1000, dropout=0.1)
        tr.set_precision("bf16")
        tr.inject_dropout_masks(k0, k1)
        logs = [tr.train(to_tb(batch)) for _ in range(2)]
        outs.append((logs, read_params(tr)))
    for a_, b_ in zip(outs[0][0], outs[1][0]):
        for k in a_:
            assert abs(a_[k] - b_[k]) <= 2e-3 * abs(b_[k]), (k, a_[k], b_[k])
    for n in outs[0][1]:
        for k in outs[0][1][n]:
            assert np.max(np.abs(outs[0][1][n][k] - outs[1][1][n][k])) <= 2.5 * 2 * 3e-4, (n, k)


@pytest.mark.parametrize("B", [1024, 600, 1000])
def test_row_kernel_column_split_is_bit_identical(B, monkeypatch):
    """Up to 1 024 rows the row kernel spreads a row tile over two blocks (each half of the dH0 columns, one shared slab
    of sums; IQLHIP_LB_CSPLIT=0 keeps one block per tile): every sum is still formed by one block in the same order, so
    gradients, losses and the parameters after two steps are bit for bit the same (ragged last tiles and an odd tile count
    included)."""
    build, read_params, to_tb, unflat = _hip()
    S, A = 39, 28
    params, batch, hyper, lrs = _case(S, A, B, seed=900 + B, gaussian=True)
    outs = []
    for cs in ("1", "0"):
        monkeypatch.setenv("IQLHIP_LB_CSPLIT", cs)
        tr = build(params, S, A, True, hyper, lrs, 1000)
        tr.set_precision("bf16")
        tb = to_tb(batch)
        flat = tr.flat_gradient(tb).copy()
        logs = [tr.train(tb) for _ in range(2)]
        outs.append((flat, logs, read_params(tr)))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1]
    for n in outs[0][2]:
        for k in outs[0][2][n]:
            assert np.array_equal(outs[0][2][n][k], outs[1][2][n][k]), (n, k)


@pytest.mark.parametrize("K,B", [(6, 256), (5, 1024), (3, 600), (3, 4096)])
def test_bf16_train_steps_graph_matches_eager_steps_on_same_indices(K, B):
    """bf16 precision: K steps through the multi-step driver (chunk graphs, device index draw, rows staged by the forward's
    idle blocks) equal K eager bf16 steps fed with the same indices — bitwise.  256 rows run the small-batch bf16 kernels,
    600 and 1 024 rows the large-batch ones (whose idle blocks also transpose W1 for the backward; row kernel: two blocks
    per tile), 4 096 rows the large-batch forward whose idle blocks first walk 3/8 of the policy's row tiles and stage
    the next step's rows behind them."""
    import iql
    import iqlhip_binding as hb
    build, read_params, _, _ = _hip()
    S, A, N = 39, 28, 4000
    params = synth.synth_params(S, A, seed=21)
    hyper = {"iql_tau": 0.8, "beta": 3.0, "discount": 0.99, "tau": 0.005}
    lrs = {"v": 3e-4, "q": 3e-4, "pi": 3e-4}
    data = synth.synth_transitions(N, S, A, seed=22)
    buf = iql.ReplayBuffer(S, A, N, "cuda")
    buf.load_d4rl_dataset({k: v.copy() for k, v in data.items()})
    g = build(params, S, A, True, hyper, lrs, 1000)
    g.set_precision("bf16")
    losses = g.train_steps(buf, K, B, seed=77)
    assert losses.shape == (K, 3) and np.all(np.isfinite(losses)) and g.total_it == K
    e = build(params, S, A, True, hyper, lrs, 1000)
    e.set_precision("bf16")
    idx = torch.empty(K * B, dtype=torch.int64, device="cuda")
    hb.check(hb.lib().iqlhip_draw_indices(idx.data_ptr(), K * B, N, 77, 0, torch.cuda.current_stream().cuda_stream))
    for k in range(K):
        log = e.train(buf.gather(idx[k * B:(k + 1) * B]))
        assert [log["value_loss"], log["q_loss"], log["actor_loss"]] == [float(x) for x in losses[k]], k
    pa, pb = read_params(g), read_params(e)
    for n in pa:
        for kk in pa[n]:
            assert np.array_equal(pa[n][kk], pb[n][kk]), (n, kk)
