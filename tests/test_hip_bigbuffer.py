"""Buffers of the size the reference's own YAMLs create (buffer_size: 10000000): 10 M rows at obs 29 / act 8
(configs/finetune/iql/antmaze/large_diverse_v2.yaml:4 -> 272-byte rows, 2.72 GB: byte offsets beyond 2^31) and at
obs 39 / act 28 (configs/offline/iql/door/human_v1.yaml:5 -> 432-byte rows, 4.32 GB: beyond 2^32).  Every kernel that
addresses rows (write, gather, packed gather, the self-drawn gather of the chunk graphs, ring write, column moments,
normalisation) is exercised on rows on BOTH sides of those lines, bit for bit against host arithmetic; and a row index
outside the buffer raises IndexError like the reference's tensor indexing (iql.py:173-177) instead of reaching a
kernel.  All indices used here are valid rows of the buffer (the round-2 fault came from a probe index that was not)."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu

N = 10_000_000
_HYPER = {"iql_tau": 0.9, "beta": 10.0, "discount": 0.99, "tau": 0.005}
_LRS = {"v": 3e-4, "q": 3e-4, "pi": 3e-4}


def _pattern(rows, width, mul, mod):
    """f32 values that are exact integers < 2^24: cell (i, j) = (i * mul + j * 17) % mod, scaled into [-1, 1)."""
    i = np.asarray(rows, dtype=np.uint32)[:, None]
    j = np.arange(width, dtype=np.uint32)[None, :]
    return ((((i * np.uint32(mul) + j * np.uint32(17)) % np.uint32(mod)).astype(np.float32)) - np.float32(mod // 2)) / np.float32(mod // 2)


def _dataset(rows, S, A):
    rows = np.asarray(rows)
    return {"observations": _pattern(rows, S, 31, 1000003), "actions": _pattern(rows, A, 37, 999983) * np.float32(0.999),
            "next_observations": _pattern(rows, S, 41, 999979), "rewards": _pattern(rows, 1, 43, 999961)[:, 0],
            "terminals": (np.asarray(rows, dtype=np.uint32) % np.uint32(97) == 0).astype(np.float32)}


def _assert_rows(batch, rows, S, A):
    want = _dataset(rows, S, A)
    s, a, r, ns, d = (t.cpu().numpy() for t in batch)
    assert np.array_equal(s, want["observations"]) and np.array_equal(a, want["actions"])
    assert np.array_equal(ns, want["next_observations"])
    assert np.array_equal(r[:, 0], want["rewards"]) and np.array_equal(d[:, 0], want["terminals"])


@pytest.mark.parametrize("S,A,lines", [(29, 8, (2 ** 31,)), (39, 28, (2 ** 31, 2 ** 32))])
def test_ten_million_row_buffer_of_the_reference_yamls(S, A, lines):
    import iql
    import iqlhip_binding as hb
    from hip_helpers import build_hip_trainer, read_params
    ld = hb.row_stride(S, A)
    row_bytes = 4 * ld
    assert N * row_bytes > max(lines), "the buffer must reach beyond the line it is meant to cross"
    n_load = N - 1                                   # one row left free: the ring write below lands on the LAST row
    data = _dataset(np.arange(n_load), S, A)
    buf = iql.ReplayBuffer(S, A, N, "cuda")
    buf.load_d4rl_dataset(data)
    assert buf._rows.shape == (N, ld) and buf._size == n_load and buf._pointer == n_load
    assert buf._rows.numel() * 4 == N * row_bytes

    # ---- rows on both sides of every line, first and last row: both gather forms, bit for bit
    probe = [0, 1, n_load - 1]
    for line in lines:
        r0 = line // row_bytes                       # the row that contains byte `line`
        assert 2 < r0 < n_load - 2
        probe += [r0 - 1, r0, r0 + 1]
    assert max(probe) < n_load and min(probe) >= 0   # valid rows only
    idx = torch.tensor(probe, dtype=torch.int64, device="cuda")
    _assert_rows(buf.gather(idx), probe, S, A)
    _assert_rows(buf.gather_split(idx), probe, S, A)
    _assert_rows(buf.gather(torch.tensor([-N + 1], dtype=torch.int64)), [1], S, A)     # negative indices wrap like torch's

    # ---- an index outside the buffer raises (it never reaches a kernel); the buffer stays usable
    for bad in ([0, N], [N + 12345], [-N - 1], [3, 2 ** 40]):
        with pytest.raises(IndexError):
            buf.gather(torch.tensor(bad, dtype=torch.int64, device="cuda"))
        with pytest.raises(IndexError):
            buf.gather_split(torch.tensor(bad, dtype=torch.int64))
    _assert_rows(buf.gather(idx), probe, S, A)

    # ---- sample(): the reference's host index draw over the whole range
    np.random.seed(5)
    want_idx = np.random.randint(0, n_load, size=512)
    np.random.seed(5)
    batch = buf.sample(512)
    assert (want_idx * row_bytes).max() > max(lines)
    _assert_rows(batch, want_idx, S, A)

    # ---- the ring write of add_transition on the last row (byte offset N * row_bytes - row_bytes), pointer wraps
    one = _dataset([n_load], S, A)
    buf.add_transition(one["observations"][0], one["actions"][0], float(one["rewards"][0]), one["next_observations"][0],
                       bool(one["terminals"][0]))
    assert buf._size == N and buf._pointer == 0
    _assert_rows(buf.gather(torch.tensor([N - 1, N - 2], dtype=torch.int64, device="cuda")), [N - 1, N - 2], S, A)

    # ---- the chunked multi-step driver (indices drawn and rows gathered inside the graphs) == eager steps on the
    # same index stream, with rows beyond the line(s) among them
    B, K = 256, 6
    params = synth.synth_params(S, A, seed=3)
    g = build_hip_trainer(params, S, A, True, _HYPER, _LRS, 1_000_000)
    e = build_hip_trainer(params, S, A, True, _HYPER, _LRS, 1_000_000)
    losses = g.train_steps(buf, K, B, seed=11)
    ix = torch.empty(K * B, dtype=torch.int64, device="cuda")
    hb.check(hb.lib().iqlhip_draw_indices(ix.data_ptr(), K * B, N, 11, 0, torch.cuda.current_stream().cuda_stream))
    ixh = ix.cpu().numpy()
    assert ixh.min() >= 0 and ixh.max() < N
    for line in lines:
        assert (ixh * row_bytes > line).any() and (ixh * row_bytes < line).any()
    for k in range(K):
        log = e.train(buf.gather(ix[k * B:(k + 1) * B]))
        assert [log["value_loss"], log["q_loss"], log["actor_loss"]] == [float(x) for x in losses[k]], k
    pa, pb = read_params(g), read_params(e)
    for net in pa:
        for t in pa[net]:
            assert np.array_equal(pa[net][t], pb[net][t]), (net, t)

    # ---- dataset ingest over all 10 M rows (64-bit row offsets in the reductions), then the in-place normalisation
    mean, std = buf.state_mean_std(1e-3)
    obs = np.concatenate([data["observations"], one["observations"]])
    m64 = obs.mean(0, dtype=np.float64)
    s64 = obs.std(0, dtype=np.float64) + 1e-3
    assert np.max(np.abs(mean - m64) / np.maximum(np.abs(m64), 1e-3)) < 1e-5 and np.max(np.abs(std - s64) / s64) < 1e-5
    buf.normalize_states_(mean, std)
    got = buf.gather(idx)
    want = _dataset(probe, S, A)
    assert np.array_equal(got[0].cpu().numpy(), (want["observations"] - mean) / std)
    assert np.array_equal(got[3].cpu().numpy(), (want["next_observations"] - mean) / std)
    assert np.array_equal(got[1].cpu().numpy(), want["actions"])
