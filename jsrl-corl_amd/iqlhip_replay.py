"""ReplayBuffer with the reference's surface (algorithms/finetune/iql.py:122-197)
on top of ONE packed row store  [s(S) | a(A) | s'(S) | r | d | pad]  (row stride
a multiple of 16 B) instead of five separate tensors, so that a sampled
transition is one contiguous 168..448-byte read instead of five 4..156-byte ones.

On a GPU device the row writes and the five-way gather of `sample` run in
libiqlhip.so (iqlhip_rows_write / iqlhip_rows_gather); the index draw stays
`np.random.randint` on the host exactly like the reference (iql.py:172), so a
seeded run draws the same index stream.  On a CPU device (no GPU in the
process: host-logic tests, config-1 plumbing) the same storage is addressed
with torch indexing; the training step itself has no CPU path.
"""
from __future__ import annotations

from typing import Dict, List

import os

import numpy as np
import torch

import iqlhip_binding as hb

TensorBatch = List[torch.Tensor]

# (data_ptr of the observations view, rows, S, A, row stride, device) of the block the last GPU sample() produced
_last_block = None
# sample(): packed blocks (and their five views) recycled per batch size; 0 = a fresh block per call
SAMPLE_RING = max(0, int(os.environ.get("IQLHIP_SAMPLE_RING", "8")))


def _is_gpu(device) -> bool:
    return torch.device(device).type == "cuda"


try:
    _raw_stream = torch._C._cuda_getCurrentRawStream          # (device_index) -> hipStream_t as int
except AttributeError:                                        # pragma: no cover - older torch
    def _raw_stream(index: int) -> int:
        return torch.cuda.current_stream(index).cuda_stream


class ReplayBuffer:
    def __init__(self, state_dim: int, action_dim: int, buffer_size: int, device: str = "cpu"):
        self._buffer_size = buffer_size
        self._pointer = 0
        self._size = 0
        self._state_dim = state_dim
        self._action_dim = action_dim
        self._device = device
        self._gpu = _is_gpu(device)
        w = 2 * state_dim + action_dim + 2
        self._ld = hb.row_stride(state_dim, action_dim) if self._gpu else (w + 3) // 4 * 4
        self._rows = torch.zeros((buffer_size, self._ld), dtype=torch.float32, device=device)
        self._rows_ptr = self._rows.data_ptr()
        pad = self._ld - w
        self._split_sizes = [state_dim, action_dim, state_dim, 1, 1] + ([pad] if pad else [])
        self._sample_ring = {}      # batch size -> [[(block, five views, block address)] * SAMPLE_RING, next slot]
        # bumped by every method that writes rows: ImplicitQLearning.train_steps may start a call on rows its previous
        # call staged ahead only while the buffer's contents are what they were then
        self._writes = 0

    # ---- views with the reference's attribute names (iql.py:134-146) ------
    @property
    def _states(self) -> torch.Tensor:
        return self._rows[:, : self._state_dim]

    @property
    def _actions(self) -> torch.Tensor:
        return self._rows[:, self._state_dim: self._state_dim + self._action_dim]

    @property
    def _next_states(self) -> torch.Tensor:
        o = self._state_dim + self._action_dim
        return self._rows[:, o: o + self._state_dim]

    @property
    def _rewards(self) -> torch.Tensor:
        o = 2 * self._state_dim + self._action_dim
        return self._rows[:, o: o + 1]

    @property
    def _dones(self) -> torch.Tensor:
        o = 2 * self._state_dim + self._action_dim + 1
        return self._rows[:, o: o + 1]

    def _to_tensor(self, data: np.ndarray) -> torch.Tensor:
        return torch.tensor(data, dtype=torch.float32, device=self._device)

    def _stream(self):
        return _raw_stream(self._rows.device.index)

    def _write_rows(self, row0: int, s, a, r, ns, d) -> None:
        n = s.shape[0]
        self._writes += 1
        if self._gpu:
            s, a, r, ns, d = (t.contiguous() for t in (s, a, r, ns, d))
            hb.check(hb.lib().iqlhip_rows_write(
                self._rows.data_ptr(), self._ld, self._state_dim, self._action_dim, row0, n,
                s.data_ptr(), a.data_ptr(), r.data_ptr(), ns.data_ptr(), d.data_ptr(), self._stream()))
        else:
            S, A = self._state_dim, self._action_dim
            blk = self._rows[row0: row0 + n]
            blk[:, :S] = s
            blk[:, S: S + A] = a
            blk[:, S + A: 2 * S + A] = ns
            blk[:, 2 * S + A] = r.reshape(-1)
            blk[:, 2 * S + A + 1] = d.reshape(-1)

    # Loads data in d4rl format, i.e. from Dict[str, np.array] (iql.py:153-169).
    def load_d4rl_dataset(self, data: Dict[str, np.ndarray]):
        if self._size != 0:
            raise ValueError("Trying to load data into non-empty replay buffer")
        n_transitions = data["observations"].shape[0]
        if n_transitions > self._buffer_size:
            raise ValueError("Replay buffer is smaller than the dataset you are trying to load!")
        chunk = 1 << 20  # bound the staging copies for 10M-row datasets
        for lo in range(0, n_transitions, chunk):
            hi = min(lo + chunk, n_transitions)
            self._write_rows(
                lo,
                self._to_tensor(data["observations"][lo:hi]),
                self._to_tensor(data["actions"][lo:hi]),
                self._to_tensor(data["rewards"][lo:hi]),
                self._to_tensor(data["next_observations"][lo:hi]),
                self._to_tensor(data["terminals"][lo:hi]),
            )
        self._size += n_transitions
        self._pointer = min(self._size, n_transitions)
        print(f"Dataset size: {n_transitions}")

    def fill_synthetic(self, n_transitions: int, seed: int = 0, p_done: float = 0.01, antmaze_rewards: bool = False) -> None:
        """Bench helper (not part of the reference's surface): fill the first n rows of an EMPTY buffer with synthetic
        D4RL-shaped transitions generated on the device (SURVEY §8d's distributions) and account for them like
        load_d4rl_dataset does (iql.py:153-169) — no host arrays, no upload; identical rows on every rank."""
        if not self._gpu:
            raise RuntimeError("iqlhip: fill_synthetic runs in libiqlhip.so and needs a GPU buffer")
        if self._size != 0:
            raise ValueError("Trying to load data into non-empty replay buffer")
        if n_transitions > self._buffer_size:
            raise ValueError("Replay buffer is smaller than the dataset you are trying to load!")
        self._writes += 1
        with torch.cuda.device(self._rows.device):
            hb.check(hb.lib().iqlhip_rows_fill_synth(self._rows.data_ptr(), self._ld, self._state_dim, self._action_dim, 0,
                                                     n_transitions, int(seed), float(p_done), int(antmaze_rewards),
                                                     self._stream()))
        self._size += n_transitions
        self._pointer = min(self._size, n_transitions)

    # ---- dataset ingest on the device (not part of the reference's surface) -------------------
    def state_mean_std(self, eps: float = 1e-3):
        """compute_mean_std (iql.py:77-80) of the stored `observations` column block, reduced on the device:
        (mean, std + eps) as float32 numpy arrays [state_dim]."""
        import iqlhip_ingest as ing
        if not self._gpu:
            raise RuntimeError("iqlhip: state_mean_std runs in libiqlhip.so and needs a GPU buffer")
        if self._size < 1:
            raise ValueError("replay buffer is empty")
        mean, std = ing.cols_mean_std(self._rows[: self._size, : self._state_dim], eps)
        return mean.cpu().numpy(), std.cpu().numpy()

    def normalize_states_(self, mean: np.ndarray, std: np.ndarray) -> None:
        """normalize_states (iql.py:83-84) applied IN PLACE to the s and s' columns of every stored row."""
        if not self._gpu:
            raise RuntimeError("iqlhip: normalize_states_ runs in libiqlhip.so and needs a GPU buffer")
        dev = self._rows.device
        m = torch.as_tensor(np.ascontiguousarray(mean, dtype=np.float32).reshape(-1)).to(dev)
        s = torch.as_tensor(np.ascontiguousarray(std, dtype=np.float32).reshape(-1)).to(dev)
        if m.numel() != self._state_dim or s.numel() != self._state_dim:
            raise ValueError("mean / std must have state_dim entries")
        self._writes += 1
        with torch.cuda.device(dev):
            hb.check(hb.lib().iqlhip_rows_normalize(self._rows.data_ptr(), self._ld, self._state_dim, self._action_dim,
                                                    0, self._size, m.data_ptr(), s.data_ptr(), self._stream()))

    def _index_bound(self) -> int:
        return self._size

    def sample_indices(self, batch_size: int) -> torch.Tensor:
        """The reference's host index draw (global numpy RNG), as a device int64 tensor."""
        indices = np.random.randint(0, self._index_bound(), size=batch_size)
        if not self._gpu:
            return torch.from_numpy(indices)
        # H2D through pinned staging buffers: a ring of 4, each guarded by an event so that a buffer is never
        # rewritten while its copy may still be queued behind earlier work on the stream
        ring = getattr(self, "_idx_ring", None)
        if ring is None or ring[0][0].shape[0] < batch_size:
            ring = [(torch.empty(max(batch_size, 256), dtype=torch.int64).pin_memory(), torch.cuda.Event())
                    for _ in range(4)]
            self._idx_ring, self._idx_slot = ring, 0
        host, done = ring[self._idx_slot]
        self._idx_slot = (self._idx_slot + 1) % len(ring)
        done.synchronize()            # no-op unless this slot's previous copy has not run yet
        host.numpy()[:batch_size] = indices
        dev = self._rows.device
        if torch.cuda.current_device() == dev.index:
            dev_idx = host[:batch_size].to(dev, non_blocking=True)
            done.record()
        else:
            with torch.cuda.device(dev):
                dev_idx = host[:batch_size].to(dev, non_blocking=True)
                done.record()
        return dev_idx

    def _checked_indices(self, idx: torch.Tensor) -> torch.Tensor:
        """The reference's `self._states[indices]` (iql.py:173-177) raises IndexError for an index outside
        [-buffer_size, buffer_size) and wraps negative ones; a gather kernel would read out of bounds.  Checked on the
        host before anything is launched (one small device reduction + read-back when the indices live on the GPU)."""
        cap = self._buffer_size
        if idx.dtype != torch.int64:
            idx = idx.to(torch.int64)
        if idx.numel() == 0:
            return idx
        lo, hi = (int(v) for v in torch.stack((idx.min(), idx.max())).tolist())
        if lo < -cap or hi >= cap:
            bad = hi if hi >= cap else lo
            raise IndexError(f"index {bad} is out of bounds for dimension 0 with size {cap}")
        if lo < 0:
            idx = torch.where(idx < 0, idx + cap, idx)
        return idx

    def gather(self, idx: torch.Tensor) -> TensorBatch:
        n = idx.shape[0]
        S, A = self._state_dim, self._action_dim
        if not self._gpu:
            rows = self._rows[idx]
            return [rows[:, :S].contiguous(), rows[:, S: S + A].contiguous(),
                    rows[:, 2 * S + A: 2 * S + A + 1].contiguous(), rows[:, S + A: 2 * S + A].contiguous(),
                    rows[:, 2 * S + A + 1: 2 * S + A + 2].contiguous()]
        # one coalesced copy of whole packed rows; the five tensors of the reference's contract are views of that
        # block (shapes as in iql.py:173-177, row stride = the packed stride) — ImplicitQLearning.train() hands
        # such a block to the library in place, with no re-packing
        idx = self._checked_indices(idx.to(self._rows.device))
        block = torch.empty((n, self._ld), dtype=torch.float32, device=self._rows.device)
        hb.check(hb.lib().iqlhip_rows_gather_packed(self._rows.data_ptr(), self._ld, self._buffer_size, idx.data_ptr(), n,
                                                    block.data_ptr(), self._stream()))
        return [block[:, :S], block[:, S: S + A], block[:, 2 * S + A: 2 * S + A + 1], block[:, S + A: 2 * S + A],
                block[:, 2 * S + A + 1: 2 * S + A + 2]]

    def gather_split(self, idx: torch.Tensor) -> TensorBatch:
        """The same sample as five separate contiguous tensors (iqlhip_rows_gather)."""
        n = idx.shape[0]
        S, A = self._state_dim, self._action_dim
        dev = self._rows.device
        idx = self._checked_indices(idx.to(dev))
        out = [torch.empty((n, S), dtype=torch.float32, device=dev),
               torch.empty((n, A), dtype=torch.float32, device=dev),
               torch.empty((n, 1), dtype=torch.float32, device=dev),
               torch.empty((n, S), dtype=torch.float32, device=dev),
               torch.empty((n, 1), dtype=torch.float32, device=dev)]
        hb.check(hb.lib().iqlhip_rows_gather(
            self._rows.data_ptr(), self._ld, self._buffer_size, S, A, idx.data_ptr(), n,
            out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), out[4].data_ptr(),
            self._stream()))
        return out

    def sample(self, batch_size: int) -> TensorBatch:
        # order: states, actions, rewards(B,1), next_states, dones(B,1)  (iql.py:171-178)
        if not self._gpu:
            return self.gather(self.sample_indices(batch_size))
        # device side in ONE library call: numpy indices -> (library's pinned ring) -> H2D copy -> packed row gather
        # (same numpy draw, same rows as gather(sample_indices(n)))
        indices = np.random.randint(0, self._index_bound(), size=batch_size)
        if indices.dtype != np.int64:
            indices = indices.astype(np.int64)
        dev = self._rows.device
        # The packed block and its five views come from a ring of SAMPLE_RING pre-built ones per batch size (allocating
        # the block and splitting it cost ~8 us of host time per call — a third of sample()).  A returned batch therefore
        # stays untouched for the next SAMPLE_RING - 1 calls of sample() with that batch size; the reference's loops use
        # a batch in the train() call that follows and drop it (algorithms/finetune/iql.py:771-773, jsrl_w_iql.py:544-548).
        # IQLHIP_SAMPLE_RING=0 restores a fresh block per call.
        ring = self._sample_ring.get(batch_size) if SAMPLE_RING > 0 else None
        if ring is None:
            n_slots = max(SAMPLE_RING, 1)
            slots = []
            for _ in range(n_slots):
                block = torch.empty((batch_size, self._ld), dtype=torch.float32, device=dev)
                # five views of the block in ONE split (s, a, s', r, d[, pad]) -> the reference's order s, a, r, s', d
                parts = block.split(self._split_sizes, dim=1)
                slots.append((block, (parts[0], parts[1], parts[3], parts[2], parts[4]), block.data_ptr()))
            ring = [slots, 0]
            if SAMPLE_RING > 0:
                self._sample_ring[batch_size] = ring
        slots, k = ring
        ring[1] = (k + 1) % len(slots)
        block, views, block_ptr = slots[k]
        if torch.cuda.current_device() == dev.index:
            hb.check(hb.lib().iqlhip_rows_sample_packed(self._rows_ptr, self._ld, self._buffer_size,
                                                        indices.ctypes.data, batch_size, block_ptr, self._stream()))
        else:
            with torch.cuda.device(dev):
                hb.check(hb.lib().iqlhip_rows_sample_packed(self._rows_ptr, self._ld, self._buffer_size,
                                                            indices.ctypes.data, batch_size, block_ptr, self._stream()))
        batch = list(views)
        # ImplicitQLearning.train recognises a batch that IS such a freshly gathered block (it is consumed in place)
        global _last_block
        _last_block = (block_ptr, batch_size, self._state_dim, self._action_dim, self._ld, dev)
        return batch

    def add_transition(self, state: np.ndarray, action: np.ndarray, reward: float, next_state: np.ndarray,
                       done: bool):
        # one packed host row, one H2D copy (the reference issues five, iql.py:189-193)
        S, A = self._state_dim, self._action_dim
        self._writes += 1
        if self._gpu:
            # pinned staging rows (ring of 4, each guarded by an event) and an asynchronous copy
            ring = getattr(self, "_row_ring", None)
            if ring is None:
                ring = [(torch.zeros(self._ld, dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(4)]
                self._row_ring, self._row_slot = ring, 0
            host, done_ev = ring[self._row_slot]
            self._row_slot = (self._row_slot + 1) % len(ring)
            done_ev.synchronize()
            row = host.numpy()
        else:
            host = None
            row = np.zeros((self._ld,), dtype=np.float32)
        row[:S] = np.asarray(state, dtype=np.float32).reshape(-1)
        row[S: S + A] = np.asarray(action, dtype=np.float32).reshape(-1)
        row[S + A: 2 * S + A] = np.asarray(next_state, dtype=np.float32).reshape(-1)
        row[2 * S + A] = np.float32(reward)
        row[2 * S + A + 1] = np.float32(done)
        if host is not None:
            dev = self._rows.device
            if torch.cuda.current_device() == dev.index:
                self._rows[self._pointer].copy_(host, non_blocking=True)
                done_ev.record()
            else:
                with torch.cuda.device(dev):
                    self._rows[self._pointer].copy_(host, non_blocking=True)
                    done_ev.record()
        else:
            self._rows[self._pointer].copy_(torch.from_numpy(row))
        self._pointer = (self._pointer + 1) % self._buffer_size
        self._size = min(self._size + 1, self._buffer_size)


class OfflineReplayBuffer(ReplayBuffer):
    """algorithms/offline/iql.py:125-184 flavour: sample bounded by min(size, pointer),
    add_transition unimplemented."""

    def _index_bound(self) -> int:
        return min(self._size, self._pointer)

    def add_transition(self, *args, **kwargs):
        raise NotImplementedError
