"""ImplicitQLearning with the reference's constructor, attributes and checkpoint
format (algorithms/finetune/iql.py:445-606) whose train() runs the whole step
— 7 MLP forwards, 3 losses, backward, 3 Adam updates, Polyak — in libiqlhip.so
(hand-written HIP for gfx950) instead of ~700 aten calls.

Ownership model: the caller constructs the nn.Modules and the
torch.optim.Adam objects and passes them in.  On a GPU device this class
re-homes every nn.Parameter's storage into ONE flat fp32 arena (layout:
iqlhip_arena_layout) and the Adam moments into two more, so the C ABI sees
three pointers; the Parameter / optimizer objects the caller holds stay the
same Python objects and stay usable (actor.act(), state_dict(), isinstance).

There is no CPU implementation of the step: with a CPU device the object can
be constructed, checkpointed and inspected (host logic), but train() raises.
"""
from __future__ import annotations

import copy
import ctypes as C
import warnings
from typing import Any, Dict, List, Optional, Tuple

import weakref

import numpy as np
import torch
import torch.nn as nn
from torch.optim.lr_scheduler import CosineAnnealingLR

import iqlhip_binding as hb
import iqlhip_dp as dp
from iqlhip_networks import (DeterministicPolicy, GaussianPolicy, LOG_STD_MAX, LOG_STD_MIN, dropout_p, register_actor_owner,
                             linear_layers)

TensorBatch = List[torch.Tensor]
EXP_ADV_MAX = 100.0
K_MAX = 1024  # steps per captured hipGraph chunk (library limit)


def _is_gpu(device) -> bool:
    return torch.device(device).type == "cuda"


try:
    _raw_stream = torch._C._cuda_getCurrentRawStream          # (device_index) -> hipStream_t as int
except AttributeError:                                        # pragma: no cover - older torch
    def _raw_stream(index: int) -> int:
        return torch.cuda.current_stream(index).cuda_stream


def _mlp_tensors(module: nn.Module) -> Dict[str, nn.Parameter]:
    lin = linear_layers(module)
    if len(lin) != 3:
        raise NotImplementedError(
            f"iqlhip kernels are built for 2 hidden layers (3 Linear layers); this network has {len(lin)}")
    return {"w0": lin[0].weight, "b0": lin[0].bias, "w1": lin[1].weight, "b1": lin[1].bias,
            "w2": lin[2].weight, "b2": lin[2].bias}


class ImplicitQLearning:
    def __init__(self, max_action: float, actor: nn.Module, actor_optimizer: torch.optim.Optimizer,
                 q_network: nn.Module, q_optimizer: torch.optim.Optimizer, v_network: nn.Module,
                 v_optimizer: torch.optim.Optimizer, iql_tau: float = 0.7, beta: float = 3.0,
                 max_steps: int = 1000000, discount: float = 0.99, tau: float = 0.005, device: str = "cpu"):
        self.max_action = max_action
        self.qf = q_network
        self.q_target = copy.deepcopy(self.qf).requires_grad_(False).to(device)
        self.vf = v_network
        self.actor = actor
        self.v_optimizer = v_optimizer
        self.q_optimizer = q_optimizer
        self.actor_optimizer = actor_optimizer
        if max_steps is not None:
            self.actor_lr_schedule = CosineAnnealingLR(self.actor_optimizer, max_steps)
        else:
            self.actor_lr_schedule = None
        self.iql_tau = iql_tau
        self.beta = beta
        self.discount = discount
        self.tau = tau

        self.total_it = 0
        self.device = device

        # data-parallel state (set by enable_data_parallel)
        self._dp_group = None
        self._dp_world = 1
        self._dp_rank = 0
        self._dp_exchange = None      # None | "rccl" | "p2p" (in-library) | "torch" (torch.distributed.all_reduce)
        self._table_cache = None
        self._ts_token = None         # (buffer, its write count) of the last train_steps call
        self._eager_next = None       # scalars of the next eager step, computed ahead (train() on a fresh sample() block)
        self._loss_out = (C.c_float * 3)()

        self._ctx = None
        self._max_batch = 0
        self._adam_t = {"v": 0, "q": 0, "pi": 0}
        self._hyper_sent = None
        self._act_bufs = None
        if _is_gpu(device):
            self._attach(max_batch=256)
            register_actor_owner(self.actor, self)

    # ------------------------------------------------------------------ arenas
    def _net_tensors(self) -> Dict[str, Dict[str, nn.Parameter]]:
        nets = {"vf": _mlp_tensors(self.vf), "q1": _mlp_tensors(self.qf.q1), "q2": _mlp_tensors(self.qf.q2),
                "pi": _mlp_tensors(self.actor.net)}
        if hasattr(self.actor, "log_std"):
            nets["pi"]["log_std"] = self.actor.log_std
        return nets

    def _target_tensors(self) -> Dict[str, Dict[str, nn.Parameter]]:
        return {"q1": _mlp_tensors(self.q_target.q1), "q2": _mlp_tensors(self.q_target.q2)}

    def _probe_dims(self) -> Tuple[int, int, bool]:
        nets = self._net_tensors()
        S = nets["vf"]["w0"].shape[1]
        A = nets["pi"]["w2"].shape[0]
        hid = nets["vf"]["w0"].shape[0]
        ok = (nets["q1"]["w0"].shape[1] == S + A and nets["q2"]["w0"].shape[1] == S + A
              and nets["pi"]["w0"].shape[1] == S and nets["vf"]["w2"].shape[0] == 1)
        for t in nets.values():
            ok = ok and t["w0"].shape[0] == hid and tuple(t["w1"].shape) == (hid, hid) and t["w2"].shape[1] == hid
        if not ok:
            raise NotImplementedError("network shapes are not the IQL TwinQ / ValueFunction / policy MLP family")
        if hid != hb.IQLHIP_HIDDEN:
            raise NotImplementedError(f"iqlhip kernels are tiled for hidden_dim={hb.IQLHIP_HIDDEN}, got {hid}")
        if dropout_p(self.vf) > 0.0 or dropout_p(self.qf) > 0.0:
            raise NotImplementedError("dropout is implemented for the actor only (the reference has none elsewhere)")
        gaussian = "log_std" in nets["pi"]
        if not gaussian and not isinstance(self.actor, DeterministicPolicy) and not hasattr(self.actor, "net"):
            raise NotImplementedError("unknown policy class")
        return S, A, gaussian

    def _segments(self, L, nets, which: str):
        """yield (parameter, offset_in_arena) for every tensor of `nets`."""
        order = {"vf": hb.NET_V, "q1": hb.NET_Q1, "q2": hb.NET_Q2, "pi": hb.NET_PI}
        for name, tensors in nets.items():
            nl = L.net[order[name]]
            shift = L.target_src if which == "target" else 0
            for key, p in tensors.items():
                yield p, int(getattr(nl, key)) - shift

    def _attach(self, max_batch: int) -> None:
        """(Re)build the flat arenas + the library context for `max_batch` rows."""
        S, A, gaussian = self._probe_dims()
        dev = torch.device(self.device)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        if not 1 <= max_batch <= 16384:
            raise ValueError(f"iqlhip: batch of {max_batch} rows is outside the library's range [1, 16384]")
        L = hb.arena_layout(S, A, gaussian, max_batch)
        old = None
        if self._ctx is not None:
            if self._dp_world > 1:
                raise RuntimeError("iqlhip: the batch size cannot grow after enable_data_parallel(); construct the "
                                   "trainer's first step (or call reserve_batch) with the largest batch first")
            old = (self._params_arena, self._target_arena, self._m_arena, self._v_arena)
        self._S, self._A, self._gaussian, self._layout = S, A, gaussian, L
        if old is None:
            self._params_arena = torch.zeros(L.n_params, dtype=torch.float32, device=dev)
            self._target_arena = torch.zeros(L.n_target, dtype=torch.float32, device=dev)
            self._m_arena = torch.zeros(L.n_params, dtype=torch.float32, device=dev)
            self._v_arena = torch.zeros(L.n_params, dtype=torch.float32, device=dev)
            self._rehome(self._net_tensors(), self._params_arena, "params")
            self._rehome(self._target_tensors(), self._target_arena, "target")
            self._absorb_opt_state()
        else:
            self._params_arena, self._target_arena, self._m_arena, self._v_arena = old
        self._dev = dev
        d = hb.Dims(S, A, hb.IQLHIP_HIDDEN, 2, hb.POLICY_GAUSSIAN if gaussian else hb.POLICY_DETERMINISTIC, max_batch)
        h = self._hyper_struct()
        ctx = C.c_void_p()
        hb.check(hb.lib().iqlhip_create(C.byref(d), C.byref(h), dev.index, C.byref(ctx)))
        if self._ctx is not None:       # carry the Philox stream positions (dropout masks, act() noise) over
            ctr = (C.c_uint64 * 2)()
            hb.check(hb.lib().iqlhip_get_counters(self._ctx, ctr))
            hb.check(hb.lib().iqlhip_set_counters(ctx, ctr))
        self._release()                 # the previous (smaller) context, only now that the new one exists
        self._ctx = ctx
        self._table_cache = None
        self._hyper_sent = self._hyper_tuple()
        self._dropout_sent = 0.0
        self._max_batch = max_batch
        if getattr(self, "_precision", "f32") == "bf16":     # survives a re-attach for a larger batch
            hb.check(hb.lib().iqlhip_set_precision(self._ctx, 1))
        hb.check(hb.lib().iqlhip_bind(self._ctx, self._params_arena.data_ptr(), self._target_arena.data_ptr(),
                                      self._m_arena.data_ptr(), self._v_arena.data_ptr()))

    def _rehome(self, nets, arena: torch.Tensor, which: str) -> None:
        with torch.no_grad():
            for p, off in self._segments(self._layout, nets, which):
                view = arena[off: off + p.numel()].view(p.shape)
                view.copy_(p.data.to(arena.device, torch.float32))
                p.data = view

    def _param_offsets(self):
        return list(self._segments(self._layout, self._net_tensors(), "params"))

    def _optimizer_of(self, p) -> torch.optim.Optimizer:
        for opt in (self.v_optimizer, self.q_optimizer, self.actor_optimizer):
            for grp in opt.param_groups:
                if any(p is q for q in grp["params"]):
                    return opt
        raise ValueError("parameter is not owned by any of the three optimizers")

    def _absorb_opt_state(self) -> None:
        """Copy existing Adam moments (e.g. after optimizer.load_state_dict) into the flat
        arenas and re-point the optimizer's state tensors at the arena views."""
        with torch.no_grad():
            for grp_name, opt in (("v", self.v_optimizer), ("q", self.q_optimizer), ("pi", self.actor_optimizer)):
                t = 0
                for st in opt.state.values():
                    if "step" in st:
                        t = max(t, int(float(st["step"])))
                self._adam_t[grp_name] = t
            for p, off in self._param_offsets():
                opt = self._optimizer_of(p)
                st = opt.state.get(p, None)
                mv = self._m_arena[off: off + p.numel()].view(p.shape)
                vv = self._v_arena[off: off + p.numel()].view(p.shape)
                if st and "exp_avg" in st:
                    mv.copy_(st["exp_avg"].to(mv.device, torch.float32))
                    vv.copy_(st["exp_avg_sq"].to(vv.device, torch.float32))
                    st["exp_avg"], st["exp_avg_sq"] = mv, vv
                else:
                    mv.zero_()
                    vv.zero_()

    def _sync_opt_state(self) -> None:
        """Make optimizer.state look like torch.optim.Adam's after `t` steps (for state_dict())."""
        if self._ctx is None:
            return
        groups = {"v": self.v_optimizer, "q": self.q_optimizer, "pi": self.actor_optimizer}
        for p, off in self._param_offsets():
            opt = self._optimizer_of(p)
            grp_name = next(k for k, o in groups.items() if o is opt)
            t = self._adam_t[grp_name]
            if t == 0 and p not in opt.state:
                continue  # torch creates Adam state lazily at the first step
            st = opt.state[p]
            st["step"] = torch.tensor(float(t), dtype=torch.float32)
            st["exp_avg"] = self._m_arena[off: off + p.numel()].view(p.shape)
            st["exp_avg_sq"] = self._v_arena[off: off + p.numel()].view(p.shape)

    def _release(self) -> None:
        if self._ctx is not None:
            hb.check(hb.lib().iqlhip_destroy(self._ctx))
            self._ctx = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ------------------------------------------------------------------ scalars
    def _hyper_tuple(self):
        return (float(self.iql_tau), float(self.beta), float(self.discount), float(self.tau))

    def _hyper_struct(self) -> hb.Hyper:
        return hb.Hyper(self.iql_tau, self.beta, self.discount, self.tau, 1.0 - self.tau, EXP_ADV_MAX,
                        LOG_STD_MIN, LOG_STD_MAX)

    def _adam_hyper(self):
        ref = None
        for opt in (self.v_optimizer, self.q_optimizer, self.actor_optimizer):
            if len(opt.param_groups) != 1:
                raise NotImplementedError("one param group per optimizer expected (reference iql.py:673-675)")
            g = opt.param_groups[0]
            if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
                raise NotImplementedError("iqlhip implements torch.optim.Adam defaults only (no weight decay/amsgrad)")
            cur = (float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]))
            if ref is None:
                ref = cur
            elif cur != ref:
                raise NotImplementedError("the three optimizers must share betas and eps")
        return ref

    def _fill_scalars(self, sc: hb.StepScalars, t: Dict[str, int], lrs: Dict[str, float], inv_batch: float) -> None:
        b1, b2, eps = self._adam_hyper()
        for i, g in enumerate(("v", "q", "pi")):
            bc1 = 1.0 - b1 ** t[g]
            bc2 = 1.0 - b2 ** t[g]
            sc.step_size[i] = lrs[g] / bc1
            sc.bc2_sqrt[i] = bc2 ** 0.5
        sc.beta2 = b2
        sc.one_minus_beta1 = 1.0 - b1
        sc.one_minus_beta2 = 1.0 - b2
        sc.eps = eps
        sc.grad_scale = 1.0
        sc.inv_batch = inv_batch

    def _current_lrs(self) -> Dict[str, float]:
        return {"v": float(self.v_optimizer.param_groups[0]["lr"]),
                "q": float(self.q_optimizer.param_groups[0]["lr"]),
                "pi": float(self.actor_optimizer.param_groups[0]["lr"])}

    def _step_schedule(self) -> None:
        if self.actor_lr_schedule is not None:
            self.actor_optimizer._opt_called = True  # the Adam step ran inside libiqlhip
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                self.actor_lr_schedule.step()

    def _require_gpu(self) -> None:
        if self._ctx is None:
            raise RuntimeError(
                "iqlhip: ImplicitQLearning.train needs a GPU device (device='cuda'); there is no CPU "
                "implementation of the step in this package")

    def reserve_batch(self, rows: int) -> None:
        """Size the library's scratch for batches of up to `rows` rows now (otherwise it grows on first use)."""
        self._require_gpu()
        if rows > self._max_batch:
            self._attach(max_batch=(rows + 255) // 256 * 256)

    def _prepare(self, rows: int) -> None:
        self._require_gpu()
        if rows > self._max_batch:
            self._attach(max_batch=(rows + 255) // 256 * 256)
        if self._hyper_tuple() != self._hyper_sent:
            h = self._hyper_struct()
            hb.check(hb.lib().iqlhip_set_hyper(self._ctx, C.byref(h)))
            self._hyper_sent = self._hyper_tuple()
        # actor dropout follows the module's mode, like nn.Dropout (active in train(), off in eval())
        p_eff = self._actor_dropout_p() if self.actor.training else 0.0
        if p_eff != self._dropout_sent:
            rank = self._dp_rank if self._dp_world > 1 else 0
            hb.check(hb.lib().iqlhip_set_dropout(self._ctx, p_eff, dp.rank_seed(torch.initial_seed(), rank)))
            self._dropout_sent = p_eff

    def _actor_dropout_p(self) -> float:
        """max p over the actor's nn.Dropout layers; the layer list is cached (walking named_modules costs ~13 us a
        step), the p values are read live."""
        mods = getattr(self, "_actor_dropouts", None)
        if mods is None or self._actor_dropouts_of is not self.actor:
            mods = [m for m in self.actor.modules() if isinstance(m, nn.Dropout)]
            self._actor_dropouts, self._actor_dropouts_of = mods, self.actor
        return float(max((m.p for m in mods), default=0.0))

    def _stream(self):
        # the raw handle of torch's CURRENT stream on our device (torch.cuda.current_stream() builds a Stream
        # object: ~4 us per call, three calls per step)
        return _raw_stream(self._dev.index)

    def _as_dev(self, t: torch.Tensor) -> torch.Tensor:
        if t.device != self._dev or t.dtype != torch.float32:
            t = t.to(self._dev, torch.float32)
        return t.contiguous()

    def _rows_of(self, t: torch.Tensor) -> Tuple[torch.Tensor, int]:
        """(tensor on our device, row stride in floats).  Row-strided views (what ReplayBuffer.sample returns: five
        views of one packed block) are passed through as they are — the library consumes such a block in place."""
        if t.device != self._dev or t.dtype != torch.float32:
            t = t.to(self._dev, torch.float32)
        if t.dim() == 2 and t.stride(0) >= t.shape[1] and (t.shape[1] == 1 or t.stride(1) == 1):
            return t, t.stride(0)
        if t.dim() == 1 and t.stride(0) >= 1:
            return t, t.stride(0)
        t = t.contiguous()
        return t, (t.shape[1] if t.dim() == 2 else 1)

    def _batch_struct(self, batch: TensorBatch):
        observations, actions, rewards, next_observations, dones = batch
        if isinstance(self.actor, DeterministicPolicy) or not self._gaussian:
            if actions.dim() != 2 or actions.shape[1] != self._A:
                raise RuntimeError("Actions shape missmatch")
        rows = [self._rows_of(x) for x in (observations, actions, rewards, next_observations, dones)]
        keep = [t for t, _ in rows]
        o, a, r, no, d = keep
        B = o.shape[0]
        if o.dim() != 2 or o.shape[1] != self._S or tuple(no.shape) != (B, self._S) or tuple(a.shape) != (B, self._A):
            raise ValueError(f"batch shapes do not match state_dim={self._S}, action_dim={self._A}")
        if r.numel() != B or d.numel() != B:
            raise ValueError("rewards / dones must have one element per row")
        b = hb.Batch(o.data_ptr(), a.data_ptr(), r.data_ptr(), no.data_ptr(), d.data_ptr(),
                     rows[0][1], rows[1][1], rows[2][1], rows[3][1], rows[4][1], None, B)
        return b, keep, B

    # ------------------------------------------------------------------ the step
    def _run_step(self, b: "hb.Batch", B: int, sync: bool):
        self.total_it += 1
        for g in self._adam_t:
            self._adam_t[g] += 1
        sc = hb.StepScalars()
        lib = hb.lib()
        if self._dp_world > 1 and self._dp_exchange == "torch":
            # the collective issued by torch.distributed between the two halves of the step (two library calls)
            self._fill_scalars(sc, self._adam_t, self._current_lrs(), dp.inv_batch(B, self._dp_world))
            flat = self._dp_flat()
            hb.check(lib.iqlhip_forward_backward(self._ctx, C.byref(b), C.byref(sc), flat.data_ptr(), self._stream()))
            dp.reduce_and_update(
                flat, lambda f: hb.check(lib.iqlhip_apply_update(self._ctx, f.data_ptr(), C.byref(sc), self._stream())),
                self._dp_group)
        elif self._dp_exchange is not None:
            # in-library exchange (RCCL all-reduce or direct peer reads) inside iqlhip_step
            self._fill_scalars(sc, self._adam_t, self._current_lrs(), dp.inv_batch(B, self._dp_world))
            hb.check(lib.iqlhip_step(self._ctx, C.byref(b), C.byref(sc), self._stream()))
        else:
            self._fill_scalars(sc, self._adam_t, self._current_lrs(), 1.0 / B)
            hb.check(lib.iqlhip_step(self._ctx, C.byref(b), C.byref(sc), self._stream()))
        self._advance_schedule(1)        # == actor_lr_schedule.step(), bit for bit, without torch's ~15 us of Python
        if not sync:
            return None
        out = (C.c_float * 3)()
        hb.check(lib.iqlhip_read_losses(self._ctx, out, self._stream()))
        return {"value_loss": float(out[0]), "q_loss": float(out[1]), "actor_loss": float(out[2])}

    def train(self, batch: TensorBatch) -> Dict[str, float]:
        """One IQL gradient step (iql.py:542-563).  Returns the three losses as floats
        (one host sync instead of the reference's three .item() calls)."""
        self._require_gpu()
        fast = self._fresh_block_batch(batch)
        if fast is not None:
            return self._train_fresh_block(*fast)
        b, keep, B = self._batch_struct(batch)
        self._prepare(B)
        log = self._run_step(b, B, sync=True)
        del keep
        return log

    # ---- the reference loop's `batch = buffer.sample(B); trainer.train(batch)` (finetune/iql.py:771-773): the batch IS
    # the block ReplayBuffer.sample just gathered — no per-tensor inspection, scalars computed while the GPU ran the
    # previous step, losses through host-mapped words the library spins on (iqlhip_step_sync)
    def _fresh_block_batch(self, batch):
        import iqlhip_replay as rp
        lb = rp._last_block
        if lb is None or self._dp_world > 1 or len(batch) != 5:
            return None
        ptr, B, S, A, ld, dev = lb
        o = batch[0]
        if o.data_ptr() != ptr or S != self._S or A != self._A or dev != self._dev or o.shape[0] != B:
            return None
        # the other four must be that block's views too (a caller may have swapped one for its own tensor)
        if (batch[1].data_ptr() != ptr + 4 * S or batch[3].data_ptr() != ptr + 4 * (S + A)
                or batch[2].data_ptr() != ptr + 4 * (2 * S + A) or batch[4].data_ptr() != ptr + 4 * (2 * S + A + 1)):
            return None
        return ptr, B, ld

    def _train_fresh_block(self, ptr: int, B: int, ld: int) -> Dict[str, float]:
        self._prepare(B)
        S, A = self._S, self._A
        b = hb.Batch(ptr, ptr + 4 * S, ptr + 4 * (2 * S + A), ptr + 4 * (S + A), ptr + 4 * (2 * S + A + 1),
                     ld, ld, ld, ld, ld, None, B)
        inv_batch = 1.0 / B
        pre = self._eager_next
        self._eager_next = None
        key = self._table_key(inv_batch)
        if pre is not None and pre[0] == key:
            sc, lr_after, sched_after = pre[1], pre[2], pre[3]
        else:
            t1 = {g: t + 1 for g, t in self._adam_t.items()}
            sc = hb.StepScalars()
            self._fill_scalars(sc, t1, self._current_lrs(), inv_batch)
            pk = self._peek_schedule(1)
            lr_after, sched_after = (None, None) if pk is None else (None, pk[1])
        lib, stream, out = hb.lib(), self._stream(), self._loss_out
        rc = lib.iqlhip_step_begin(self._ctx, C.byref(b), C.byref(sc), stream)
        if rc:
            hb.check(rc)
        # ---- the GPU runs the step: host bookkeeping and the NEXT step's scalars (float64 Adam bias corrections, the
        # cosine learning rate) now; a set computed ahead is used only if nothing it depends on has changed by then
        self.total_it += 1
        for g in self._adam_t:
            self._adam_t[g] += 1
        if sched_after is not None:
            self._commit_schedule(sched_after)
        else:
            self._advance_schedule(1)
        pk = self._peek_schedule(1)
        if pk is not None:
            t2 = {g: t + 1 for g, t in self._adam_t.items()}
            sc2 = hb.StepScalars()
            self._fill_scalars(sc2, t2, self._current_lrs(), inv_batch)
            self._eager_next = (self._table_key(inv_batch), sc2, None, pk[1])
        rc = lib.iqlhip_step_wait(self._ctx, out, stream)
        if rc:
            hb.check(rc)
        log = {"value_loss": float(out[0]), "q_loss": float(out[1]), "actor_loss": float(out[2])}
        return log

    def online_step(self, replay_buffer, state, action, reward: float, next_state, done: bool,
                    batch_size: int, act_next: Optional[np.ndarray] = None):
        """One iteration of the online loop's buffer + training work in ONE library call (reference sequence:
        `replay_buffer.add_transition(...)`, `batch = replay_buffer.sample(batch_size)`, `trainer.train(batch)` —
        algorithms/finetune/iql.py:741-773, jsrl_w_iql.py:512-548): the transition is stored at the ring pointer, the
        batch indices are drawn by np.random.randint over the NEW size (same global-RNG draw as sample()), the rows
        are gathered and the step runs; returns train()'s dict.  Equivalent to the three calls, bit for bit.
        `act_next` (a state, normally `next_state` unless the episode ended) additionally returns
        `self.actor.act(act_next, device)` evaluated with the UPDATED policy — what the loop's next iteration would
        compute first (iql.py:728) — under the same single synchronisation: returns (log, action)."""
        self._prepare(batch_size)
        buf = replay_buffer
        if not getattr(buf, "_gpu", False) or buf._rows.device != self._dev:
            raise ValueError("online_step needs a ReplayBuffer on the trainer's GPU")
        if self._dp_world > 1 and self._dp_exchange == "torch":
            raise NotImplementedError("online_step runs the whole step in one library call: it needs an in-library "
                                      "exchange, enable_data_parallel(exchange='rccl'|'p2p') — with exchange='torch' "
                                      "use add_transition / sample / train")
        S, A = self._S, self._A
        row = getattr(self, "_on_row", None)
        if row is None or row.shape[0] != buf._ld:
            row = self._on_row = np.zeros(buf._ld, dtype=np.float32)
        row[:S] = np.asarray(state, dtype=np.float32).reshape(-1)
        row[S: S + A] = np.asarray(action, dtype=np.float32).reshape(-1)
        row[S + A: 2 * S + A] = np.asarray(next_state, dtype=np.float32).reshape(-1)
        row[2 * S + A] = np.float32(reward)
        row[2 * S + A + 1] = np.float32(done)
        # (the buffer's and the trainer's counters move only once the library call has succeeded)
        pointer = buf._pointer
        new_size = min(buf._size + 1, buf._buffer_size)
        from iqlhip_replay import ReplayBuffer
        if type(buf)._index_bound is not ReplayBuffer._index_bound or type(buf).add_transition is not ReplayBuffer.add_transition:
            raise NotImplementedError("online_step needs the finetune ReplayBuffer (the offline flavour has no add_transition)")
        idx = np.random.randint(0, new_size, size=batch_size)        # sample()'s draw over the size AFTER the insert
        if idx.dtype != np.int64:
            idx = idx.astype(np.int64)
        adam_next = {g: t + 1 for g, t in self._adam_t.items()}
        sc = hb.StepScalars()
        self._fill_scalars(sc, adam_next, self._current_lrs(), dp.inv_batch(batch_size, self._dp_world))
        out = (C.c_float * 3)()
        a_in = a_out = None
        seed = 0
        if act_next is not None:
            from iqlhip_networks import dropout_p
            if self.actor.training and dropout_p(self.actor) > 0.0:
                raise NotImplementedError("act_next: the library's inference forward is eval-mode (no actor dropout)")
            a_in = np.ascontiguousarray(np.asarray(act_next, dtype=np.float32).reshape(-1))
            a_out = np.empty(A, dtype=np.float32)
            seed = self._act_seed() if (self.actor.training and self._gaussian) else 0
        hb.check(hb.lib().iqlhip_online_step(self._ctx, buf._rows.data_ptr(), buf._ld, buf._buffer_size, pointer,
                                             row.ctypes.data, idx.ctypes.data, batch_size, C.byref(sc), out,
                                             None if a_in is None else a_in.ctypes.data, float(self.actor.max_action),
                                             seed, None if a_out is None else a_out.ctypes.data, self._stream()))
        buf._writes += 1
        buf._pointer = (pointer + 1) % buf._buffer_size
        buf._size = new_size
        self.total_it += 1
        self._adam_t = adam_next
        self._advance_schedule(1)
        log = {"value_loss": float(out[0]), "q_loss": float(out[1]), "actor_loss": float(out[2])}
        return log if act_next is None else (log, a_out)

    def _schedule_state(self):
        sch = self.actor_lr_schedule
        lr = float(self.actor_optimizer.param_groups[0]["lr"])
        if sch is None:
            return (lr, None, None)
        return (lr, sch.last_epoch, sch._step_count)

    def _peek_schedule(self, k: int):
        """(actor learning rates USED by the next k steps, scheduler state after them) without touching the
        CosineAnnealingLR object.  Same float64 recursion as torch's CosineAnnealingLR.get_lr (eta_min = 0), run as a
        plain loop instead of k scheduler.step() calls (~15 us each).  None when the scheduler is in a state only
        torch should advance."""
        lr = float(self.actor_optimizer.param_groups[0]["lr"])
        out = np.empty(k, dtype=np.float64)
        sch = self.actor_lr_schedule
        if sch is None:
            out[:] = lr
            return out, (lr, None, None)
        import math
        T = sch.T_max
        base = float(sch.base_lrs[0])
        eta_min = float(sch.eta_min)
        if eta_min != 0.0 or len(sch.base_lrs) != 1 or (sch._step_count == 1 and sch.last_epoch > 0):
            return None
        e = sch.last_epoch
        cos, pi = math.cos, math.pi
        for i in range(k):
            out[i] = lr
            e += 1
            if (e - 1 - T) % (2 * T) == 0:
                lr = lr + base * (1 - cos(pi / T)) / 2
            else:
                lr = (1 + cos(pi * e / T)) / (1 + cos(pi * (e - 1) / T)) * lr
        return out, (lr, e, sch._step_count + k)

    def _commit_schedule(self, state) -> None:
        lr, e, count = state
        sch = self.actor_lr_schedule
        if sch is None:
            return
        sch.last_epoch = e
        sch._step_count = count
        self.actor_optimizer.param_groups[0]["lr"] = lr
        sch._last_lr = [lr]
        self.actor_optimizer._opt_called = True

    def _advance_schedule(self, k: int) -> np.ndarray:
        """Actor learning rates USED by the next k steps; advances the CosineAnnealingLR object by k steps."""
        pk = self._peek_schedule(k)
        if pk is None:
            out = np.empty(k, dtype=np.float64)
            for i in range(k):     # unusual scheduler state: let torch do it
                out[i] = float(self.actor_optimizer.param_groups[0]["lr"])
                self._step_schedule()
            return out
        out, state = pk
        self._commit_schedule(state)
        return out

    def _build_table(self, k: int, inv_batch: float, adam_t: Dict[str, int], lr_pi: np.ndarray) -> np.ndarray:
        """[k,12] float32 rows laid out like iqlhip_step_scalars for k steps after Adam step counts `adam_t`."""
        b1, b2, eps = self._adam_hyper()
        lrs = self._current_lrs()
        tab = np.empty((k, 12), dtype=np.float32)
        steps = np.arange(1, k + 1, dtype=np.float64)
        for i, g in enumerate(("v", "q", "pi")):
            t = adam_t[g] + steps
            lr = lr_pi if g == "pi" else lrs[g]
            tab[:, i] = lr / (1.0 - np.power(b1, t))
            tab[:, 3 + i] = np.sqrt(1.0 - np.power(b2, t))
        tab[:, 6] = b2
        tab[:, 7] = 1.0 - b1
        tab[:, 8] = 1.0 - b2
        tab[:, 9] = eps
        tab[:, 10] = 1.0
        tab[:, 11] = inv_batch
        return tab

    def _table_key(self, inv_batch: float):
        """Everything a precomputed scalar table depends on (plain attribute reads: this runs on the critical path of
        every train_steps call, in front of the first launch)."""
        t = self._adam_t
        gv, gq, gp = (self.v_optimizer.param_groups[0], self.q_optimizer.param_groups[0],
                      self.actor_optimizer.param_groups[0])
        sch = self.actor_lr_schedule
        return (inv_batch, t["v"], t["q"], t["pi"], gv["lr"], gq["lr"], gp["lr"],
                None if sch is None else (sch.last_epoch, sch._step_count, sch.T_max),
                gv["betas"], gv["eps"], gq["betas"], gq["eps"], gp["betas"], gp["eps"])

    def _scalar_table(self, k: int, inv_batch: float) -> np.ndarray:
        """The per-step scalars of the next k steps (float64 Adam bias corrections and cosine learning rates, cast to
        float32 once per step like torch does); advances the Adam step counts and the scheduler.  A table computed
        ahead of time by _lookahead_table (while the GPU was busy with the previous chunk) is used when the state it
        was computed for is still the current one."""
        key = self._table_key(inv_batch)
        cached = self._table_cache
        self._table_cache = None
        if cached is not None and cached[0] == key and cached[1].shape[0] >= k:
            # rows depend on the absolute step only: the first k rows of a longer look-ahead are this call's table
            # (passed as it is: the library copies the first k rows before it returns)
            _, tab, lr_after, (e0, c0) = cached
            self._commit_schedule((float(lr_after[k - 1]), None if e0 is None else e0 + k, None if c0 is None else c0 + k))
        else:
            lr_pi = self._advance_schedule(k)
            tab = self._build_table(k, inv_batch, self._adam_t, lr_pi)
        for g in self._adam_t:
            self._adam_t[g] += k
        return tab

    def _lookahead_table(self, k: int, inv_batch: float) -> None:
        """Precompute the table of the steps AFTER the ones just launched (host work overlapped with the GPU): at least
        64 rows, so that a following call of any length up to that finds its rows ready."""
        k = max(int(k), 64)
        pk = self._peek_schedule(k + 1)
        if pk is None:
            return
        lr_used, _ = pk                     # lr_used[i] = learning rate step i uses = the rate AFTER i steps
        sch = self.actor_lr_schedule
        state0 = (None, None) if sch is None else (sch.last_epoch, sch._step_count)
        tab = self._build_table(k, inv_batch, self._adam_t, lr_used[:k])       # (_adam_hyper inside: raises on optimizer
        self._table_cache = (self._table_key(inv_batch), tab, lr_used[1:], state0)  #  settings the kernels do not implement)

    def _train_steps_args(self, replay_buffer, batch_size: int):
        self._prepare(batch_size)
        if self._dp_world > 1 and self._dp_exchange == "torch":
            raise NotImplementedError("train_steps needs an in-library exchange: enable_data_parallel(exchange='rccl'|'p2p')")
        if not getattr(replay_buffer, "_gpu", False):
            raise ValueError("train_steps needs a ReplayBuffer that lives on the GPU")
        size = replay_buffer._index_bound()
        if size < 1:
            raise ValueError("replay buffer is empty")
        return size, dp.inv_batch(batch_size, self._dp_world)

    def prepare_train_steps(self, replay_buffer, batch_size: int) -> None:
        """Capture and upload the hipGraph chunk train_steps replays for this buffer / batch size now, so that no later
        train_steps call pays for it (it is captured lazily otherwise, by the first call of >= 64 steps)."""
        _, inv_batch = self._train_steps_args(replay_buffer, batch_size)
        hb.check(hb.lib().iqlhip_train_steps_prepare(self._ctx, replay_buffer._rows.data_ptr(), replay_buffer._ld,
                                                      batch_size, inv_batch, self._stream()))
        self._ts_token = None

    def train_steps(self, replay_buffer, n_steps: int, batch_size: int, seed: int = 0,
                    return_losses: bool = True, chunk: int = K_MAX) -> Optional[np.ndarray]:
        """n_steps consecutive `sample -> train` iterations without host round trips
        (the offline loop body, algorithms/offline/iql.py:631-635): indices are drawn on
        the device (uniform with replacement, Philox keyed by (seed, total_it)); the library
        launches the call's first 2 or 4 steps directly and replays fixed chunk graphs of 64 /
        16 / 4 / 2 / 1 steps for the rest (nothing is captured per value of n_steps); per-step
        Adam / cosine-LR scalars are precomputed on the host (the next call's while the GPU
        runs this one's).  Under data parallelism every rank draws its own rows (rank-offset
        stream) and the gradient exchange runs inside the same stream / graph.  Returns losses
        [n_steps,3] (value, q, actor) when return_losses, else None (fully asynchronous)."""
        size, inv_batch = self._train_steps_args(replay_buffer, batch_size)
        chunk = max(1, min(int(chunk), K_MAX))
        lib = hb.lib()
        rank = self._dp_rank if self._dp_world > 1 else 0
        seed = dp.rank_seed(seed, rank)
        losses = np.empty((n_steps, 3), dtype=np.float32) if return_losses else None
        rows_ptr, ld, stream = replay_buffer._rows.data_ptr(), replay_buffer._ld, self._stream()
        half = (batch_size + 1) // 2
        done = 0
        while done < n_steps:
            k = min(chunk, n_steps - done)
            tab = self._scalar_table(k, inv_batch)
            # IQLHIP_TS_CONTINUE: nothing has written this buffer's rows since our previous call on it (the buffer
            # counts its writes); the library itself checks that this call starts where that one's index stream ended
            # (a weak reference: the token must not keep a multi-GB buffer alive; the tensor's version counter sees in-place
            #  writes through the reference-named views _states / _rewards / ..., which alias the packed rows)
            tok = (weakref.ref(replay_buffer), replay_buffer._writes, replay_buffer._rows._version)
            flags = hb.TS_CONTINUE if self._ts_token == tok else 0
            rc = lib.iqlhip_train_steps(self._ctx, rows_ptr, ld, size, batch_size, tab.ctypes.data, k,
                                        seed, self.total_it * half, flags, stream)
            if rc:
                self._ts_token = None
                hb.check(rc)
            self._ts_token = tok
            self.total_it += k
            done += k
            # the GPU is busy with these k steps: compute the scalars of the next k now
            self._lookahead_table(min(chunk, n_steps - done) if done < n_steps else k, inv_batch)
            if return_losses:
                buf = (C.c_float * (3 * k))()
                hb.check(lib.iqlhip_read_loss_ring(self._ctx, buf, k, stream))
                losses[done - k: done] = np.frombuffer(buf, dtype=np.float32).reshape(k, 3)
        return losses

    def train_steps_dp(self, replay_buffer, n_steps: int, batch_size: int, seed: int = 0) -> None:
        """Data-parallel multi-step run without host syncs (= train_steps(..., return_losses=False))."""
        self.train_steps(replay_buffer, n_steps, batch_size, seed=seed, return_losses=False)

    def train_on_buffer(self, replay_buffer, batch_size: int, seed: int = 0, sync: bool = False):
        """One step on rows drawn ON THE DEVICE from `replay_buffer` (no host index draw, no
        host sync unless `sync`).  Data-parallel aware: each rank draws its own rows (seed is
        offset by the rank), gradients are all-reduced before the update."""
        self._prepare(batch_size)
        size = replay_buffer._index_bound()
        lib = hb.lib()
        idx = getattr(self, "_idx_buf", None)
        if idx is None or idx.numel() != batch_size:
            idx = torch.empty(batch_size, dtype=torch.int64, device=self._dev)
            self._idx_buf = idx
        rank = self._dp_rank if self._dp_world > 1 else 0
        hb.check(lib.iqlhip_draw_indices(idx.data_ptr(), batch_size, size, dp.rank_seed(seed, rank),
                                         int(self.total_it) * ((batch_size + 1) // 2), self._stream()))
        S, A = self._S, self._A
        base = replay_buffer._rows.data_ptr()
        ld = replay_buffer._ld
        b = hb.Batch(base, base + 4 * S, base + 4 * (2 * S + A), base + 4 * (S + A), base + 4 * (2 * S + A + 1),
                     ld, ld, ld, ld, ld, idx.data_ptr(), batch_size)
        return self._run_step(b, batch_size, sync)

    # ------------------------------------------------------------------ data parallel
    def enable_data_parallel(self, process_group=None, exchange: str = "rccl", timeout_ms: int = 5000) -> None:
        """One process per GPU: parameters replicated (rank 0's are broadcast once), each rank trains
        on its own rows, the flat gradient (+3 loss words) is summed over the ranks between the backward and the
        fused Adam/Polyak launch.  `exchange`:
          "rccl"  ncclAllReduce issued by the library in-stream (and inside the captured chunk graphs);
          "p2p"   the ranks' gradient buffers are mapped into each other (hipIpc) and the update kernel reads them
                  directly over xGMI after a flag handshake — no collective call per step (one node, <= 8 ranks);
          "both"  attach both (select with select_exchange(); "p2p" is selected);
          "torch" torch.distributed.all_reduce between two library calls per step (any backend; eager train() only).
        Call after torch.distributed is up; batch sizes must not grow afterwards (reserve_batch first)."""
        import torch.distributed as dist
        self._require_gpu()
        if exchange not in ("rccl", "p2p", "both", "torch"):
            raise ValueError("exchange must be 'rccl', 'p2p', 'both' or 'torch'")
        self._dp_group = process_group
        self._dp_world = dist.get_world_size(process_group)
        self._dp_rank = dist.get_rank(process_group)
        world, rank = self._dp_world, self._dp_rank
        if world > 1 and not (exchange == "p2p" and dist.get_backend(process_group) == "gloo"):
            dp.broadcast_state((self._params_arena, self._target_arena, self._m_arena, self._v_arena), process_group,
                               src=dist.get_global_rank(process_group, 0) if process_group else 0)
        elif world > 1:
            dp.broadcast_state_host((self._params_arena, self._target_arena, self._m_arena, self._v_arena), process_group)
        lib = hb.lib()
        if exchange == "torch":
            self._dp_exchange = "torch"
            return
        src = dist.get_global_rank(process_group, 0) if process_group else 0
        if exchange in ("rccl", "both"):
            # collective-safe like the peer mapping below: a failure on any rank (no librccl, communicator init) is
            # agreed on by all ranks before anyone relies on the collective
            err = None
            uid = (C.c_char * hb.IQLHIP_UNIQUE_ID_BYTES)()
            if rank == 0:
                try:
                    hb.check(lib.iqlhip_comm_unique_id(uid))
                except Exception as e:      # noqa: BLE001 - reported after the collective
                    err = e
            box = [None if err is not None else bytes(uid.raw)]
            dist.broadcast_object_list(box, src=src, group=process_group)
            if box[0] is None:
                err = err or RuntimeError("rank 0 could not create a communicator id")
            else:
                ubuf = C.create_string_buffer(box[0], hb.IQLHIP_UNIQUE_ID_BYTES)
                try:
                    hb.check(lib.iqlhip_allreduce_init(self._ctx, ubuf, rank, world))
                except Exception as e:      # noqa: BLE001
                    err = e
            oks = [None] * world
            dist.all_gather_object(oks, err is None, group=process_group)
            if all(oks):
                self._dp_exchange = "rccl"
            else:
                self._rccl_error = f"RCCL exchange unavailable on ranks {[i for i, o in enumerate(oks) if not o]}: {err}"
                if exchange == "rccl":
                    raise RuntimeError("iqlhip: " + self._rccl_error)
        if exchange in ("p2p", "both"):
            if world > hb.IQLHIP_MAX_WORLD:
                raise ValueError(f"the p2p exchange serves one node (<= {hb.IQLHIP_MAX_WORLD} ranks), got {world}")
            # collective-safe: every rank exports, gathers and TRIES to map its peers; the outcome is agreed on before
            # anyone relies on it, so a rank whose mapping failed (no peer access between two GPUs) cannot leave the
            # others waiting for its flags
            err = None
            h = (C.c_char * hb.IQLHIP_IPC_HANDLE_BYTES)()
            try:
                hb.check(lib.iqlhip_p2p_export(self._ctx, h, rank, world))
            except Exception as e:          # noqa: BLE001 - reported below, after the collective
                err = e
            handles = [None] * world
            dist.all_gather_object(handles, None if err is not None else bytes(h.raw), group=process_group)
            if err is None and all(x is not None for x in handles):
                blob = C.create_string_buffer(b"".join(handles), world * hb.IQLHIP_IPC_HANDLE_BYTES)
                try:
                    hb.check(lib.iqlhip_p2p_attach(self._ctx, blob, int(timeout_ms)))
                except Exception as e:      # noqa: BLE001
                    err = e
            oks = [None] * world
            dist.all_gather_object(oks, err is None, group=process_group)   # (doubles as the barrier: all blocks mapped)
            if all(oks):
                self._dp_exchange = "p2p"
            else:
                self._p2p_error = f"p2p exchange unavailable on ranks {[i for i, o in enumerate(oks) if not o]}: {err}"
                if self._dp_exchange == "rccl":          # "both": keep the collective library's exchange
                    hb.check(lib.iqlhip_xch_select(self._ctx, hb.XCH_RCCL))
                elif exchange == "p2p":
                    raise RuntimeError("iqlhip: " + self._p2p_error)
        if exchange == "both" and self._dp_exchange not in ("rccl", "p2p"):
            # neither in-library exchange came up: the eager torch.distributed exchange still trains correctly
            # (train() / train_on_buffer(); train_steps needs an in-library exchange)
            self._dp_exchange = "torch"

    def resync_replicas(self) -> None:
        """Collective: make rank 0's state the common one again — the four arenas, the step counters and the actor's
        schedule — and forget a recorded exchange timeout.  For a caller that found the replicas diverged (a probe of
        an exchange that does not work on this machine) and is about to continue on another exchange."""
        import torch.distributed as dist
        self._require_gpu()
        torch.cuda.synchronize(self._dev)
        g = self._dp_group
        src = dist.get_global_rank(g, 0) if g else 0
        if dist.get_backend(g) == "gloo":
            dp.broadcast_state_host((self._params_arena, self._target_arena, self._m_arena, self._v_arena), g, src)
        else:
            dp.broadcast_state((self._params_arena, self._target_arena, self._m_arena, self._v_arena), g, src)
        box = [(self.total_it, dict(self._adam_t), self._schedule_state())] if self._dp_rank == 0 else [None]
        dist.broadcast_object_list(box, src=src, group=g)
        total_it, adam_t, (lr, e, count) = box[0]
        self.total_it, self._adam_t = int(total_it), dict(adam_t)
        if self.actor_lr_schedule is not None:
            self._commit_schedule((lr, e, count))
        else:
            self.actor_optimizer.param_groups[0]["lr"] = lr
        self._table_cache = None
        self._ts_token = None
        hb.check(hb.lib().iqlhip_xch_clear_status(self._ctx, self._stream()))

    def select_exchange(self, exchange: str) -> None:
        """Switch between attached in-library exchanges ("rccl" / "p2p"); collective: every rank must do the same."""
        if exchange == "torch":           # the eager fallback: local steps in the library, torch.distributed in between
            hb.check(hb.lib().iqlhip_xch_select(self._ctx, hb.XCH_NONE))
            self._dp_exchange = "torch"
            return
        mode = {"rccl": hb.XCH_RCCL, "p2p": hb.XCH_P2P}[exchange]
        hb.check(hb.lib().iqlhip_xch_select(self._ctx, mode))
        self._dp_exchange = exchange

    def exchange_status(self) -> Dict[str, int]:
        """{"mode", "timed_out_step" (0 = no P2P wait ever timed out), "steps"}; synchronises the stream."""
        st = (C.c_int64 * 3)()
        hb.check(hb.lib().iqlhip_xch_status(self._ctx, st, self._stream()))
        return {"mode": int(st[0]), "timed_out_step": int(st[1]), "steps": int(st[2])}

    def _dp_flat(self) -> torch.Tensor:
        n = int(hb.lib().iqlhip_grad_words(self._ctx))
        f = getattr(self, "_dp_flat_buf", None)
        if f is None or f.numel() != n:
            f = torch.zeros(n, dtype=torch.float32, device=self._dev)
            self._dp_flat_buf = f
        return f

    # ------------------------------------------------------------------ checkpoints
    def state_dict(self) -> Dict[str, Any]:
        self._sync_opt_state()
        if self.actor_lr_schedule is None:
            lr_state_dict = {}
        else:
            lr_state_dict = self.actor_lr_schedule.state_dict()
        return {
            "qf": self.qf.state_dict(),
            "q_optimizer": self.q_optimizer.state_dict(),
            "vf": self.vf.state_dict(),
            "v_optimizer": self.v_optimizer.state_dict(),
            "actor": self.actor.state_dict(),
            "actor_optimizer": self.actor_optimizer.state_dict(),
            "actor_lr_schedule": lr_state_dict,
            "total_it": self.total_it,
        }

    def _reset_target_from_qf(self) -> None:
        # reference: q_target = copy.deepcopy(qf) WITHOUT requires_grad_(False) (iql.py:584, Appendix A quirk)
        if self._ctx is None:
            self.q_target = copy.deepcopy(self.qf)
            return
        L = self._layout
        with torch.no_grad():
            self._target_arena.copy_(self._params_arena[L.target_src: L.target_src + L.n_target])
        new_t = copy.deepcopy(self.qf)   # fresh module object, like the reference
        self.q_target = new_t
        self._rehome(self._target_tensors(), self._target_arena, "target")

    def load_state_dict(self, state_dict: Dict[str, Any]):
        self.qf.load_state_dict(state_dict["qf"])
        self.q_optimizer.load_state_dict(state_dict["q_optimizer"])
        self._reset_target_from_qf()

        self.vf.load_state_dict(state_dict["vf"])
        self.v_optimizer.load_state_dict(state_dict["v_optimizer"])
        self.actor.load_state_dict(state_dict["actor"])
        self.actor_optimizer.load_state_dict(state_dict["actor_optimizer"])
        if self.actor_lr_schedule is not None:
            self.actor_lr_schedule.load_state_dict(state_dict["actor_lr_schedule"])

        self.total_it = state_dict["total_it"]
        if self._ctx is not None:
            self._absorb_opt_state()

    def partial_load_state_dict(self, state_dict: Dict[str, Any]):
        """Load state dict, but don't load optimisers (iql.py:595-606)."""
        self.qf.load_state_dict(state_dict["qf"])
        self._reset_target_from_qf()

        self.vf.load_state_dict(state_dict["vf"])

        self.actor.load_state_dict(state_dict["actor"])
        if self.actor_lr_schedule is not None:
            self.actor_lr_schedule.load_state_dict(state_dict["actor_lr_schedule"])

        self.total_it = state_dict["total_it"]

    # ------------------------------------------------------------------ introspection for tests / bench
    # ------------------------------------------------------------------ policy inference
    def can_act_on(self, device) -> bool:
        """True if the library context can serve actor.act(state, device) (same GPU as the arenas)."""
        if self._ctx is None or not _is_gpu(device):
            return False
        d = torch.device(device)
        return d.index is None or d.index == self._dev.index

    def actor_forward(self, states: torch.Tensor, sample: bool = False, max_action: Optional[float] = None) -> torch.Tensor:
        """Batched policy forward on the device: [n, S] float32 -> actions [n, A]
        = clamp(max_action * (tanh(MLP(s)) [+ sigma * N(0,1) if `sample` and the policy is Gaussian]), +-max_action).
        The batched form of GaussianPolicy.act / DeterministicPolicy.act (iql.py:371-379, 404-413) for evaluation
        loops; eval-mode forward (no dropout)."""
        self._require_gpu()
        x = self._as_dev(states).reshape(-1, self._S)
        n = x.shape[0]
        ma = float(self.max_action if max_action is None else max_action)
        out = torch.empty((n, self._A), dtype=torch.float32, device=self._dev)
        draw = sample and self._gaussian          # dist.sample(): noise drawn on the device, seeded from torch's seed
        st = self._stream()
        cap = max(self._max_batch, hb.IQLHIP_ACT_ROWS)
        for r0 in range(0, n, cap):
            r1 = min(n, r0 + cap)
            if draw:
                hb.check(hb.lib().iqlhip_actor_sample(self._ctx, x[r0:r1].data_ptr(), x.stride(0), r1 - r0,
                                                      self._act_seed(), ma, out[r0:r1].data_ptr(), out.stride(0), st))
            else:
                hb.check(hb.lib().iqlhip_actor_forward(self._ctx, x[r0:r1].data_ptr(), x.stride(0), r1 - r0, None,
                                                       self._A, ma, out[r0:r1].data_ptr(), out.stride(0), st))
        return out

    def _act_seed(self) -> int:
        """Non-zero 64-bit key of the device noise stream of act(): torch's seed at first use (so that
        torch.manual_seed(s) before training gives a reproducible exploration stream)."""
        k = getattr(self, "_act_key", None)
        if k is None:
            k = (int(torch.initial_seed()) * 0x9E3779B97F4A7C15 + 0xAC7) & 0xFFFFFFFFFFFFFFFF
            self._act_key = k = (k or 1)
        return k

    def act_one(self, state: np.ndarray, max_action: float, sample: bool) -> np.ndarray:
        """actor.act(state): one state in, one action out.  The state is written into a pinned host buffer that the
        pack kernel reads directly and the action lands in a pinned host buffer the finish kernel writes directly
        (host-mapped memory over PCIe: ~120 B each way), so the call is three launches and one stream
        synchronisation — no staging copies, no per-call allocation."""
        self._require_gpu()
        if self._act_bufs is None:
            h_in = torch.empty((1, self._S), dtype=torch.float32).pin_memory()
            h_out = torch.empty((1, self._A), dtype=torch.float32).pin_memory()
            self._act_bufs = (h_in, h_out, h_in.numpy(), h_out.numpy())
        h_in, h_out, in_np, out_np = self._act_bufs
        in_np[0, :] = np.asarray(state, dtype=np.float32).reshape(-1)
        st = self._stream()
        if sample and self._gaussian:
            hb.check(hb.lib().iqlhip_actor_sample(self._ctx, h_in.data_ptr(), self._S, 1, self._act_seed(),
                                                  float(max_action), h_out.data_ptr(), self._A, st))
        else:
            hb.check(hb.lib().iqlhip_actor_forward(self._ctx, h_in.data_ptr(), self._S, 1, None, self._A,
                                                   float(max_action), h_out.data_ptr(), self._A, st))
        hb.check(hb.lib().iqlhip_stream_synchronize(st))
        return out_np[0].copy()

    def set_timing(self, enabled: bool) -> None:
        self._require_gpu()
        hb.check(hb.lib().iqlhip_set_timing(self._ctx, 1 if enabled else 0))

    def get_timing_us(self):
        out = (C.c_float * 4)()
        hb.check(hb.lib().iqlhip_get_timing(self._ctx, out))
        return [float(x) for x in out]

    def time_kernel(self, batch: TensorBatch, which: int, repeat: int = 200) -> float:
        """Average microseconds per launch of one kernel of the step, launched back to back."""
        b, keep, B = self._batch_struct(batch)
        self._prepare(B)
        out = C.c_float(0)
        hb.check(hb.lib().iqlhip_debug_time_kernel(self._ctx, C.byref(b), which, repeat, C.byref(out), self._stream()))
        return float(out.value)

    def set_precision(self, mode: str) -> None:
        """"f32" (default, the parity path) or "bf16": bf16 operands / fp32 accumulate for the layer
        and weight-gradient products (BASELINE config 5's MFMA bf16 path; heads, losses, Adam stay fp32).  Not part of the reference's surface."""
        self._require_gpu()
        if mode not in ("f32", "bf16"):
            raise ValueError("precision must be 'f32' or 'bf16'")
        self._precision = mode
        hb.check(hb.lib().iqlhip_set_precision(self._ctx, 1 if mode == "bf16" else 0))

    def inject_dropout_masks(self, keep0: np.ndarray, keep1: np.ndarray) -> None:
        """Tests: use these keep-masks (bool [rows,256] per layer) for the following steps instead of device draws."""
        from synth import pack_keep_bits
        self._prepare(keep0.shape[0])
        b0 = np.ascontiguousarray(pack_keep_bits(keep0))
        b1 = np.ascontiguousarray(pack_keep_bits(keep1))
        hb.check(hb.lib().iqlhip_debug_write_masks(self._ctx, b0.ctypes.data, b1.ctypes.data, keep0.shape[0],
                                                   self._stream()))

    def debug_read(self, name: str) -> np.ndarray:
        self._require_gpu()
        cap = 4 * self._max_batch * 256 + 64
        buf = (C.c_float * cap)()
        n = C.c_int64(0)
        hb.check(hb.lib().iqlhip_debug_read(self._ctx, name.encode(), buf, cap, C.byref(n), self._stream()))
        return np.frombuffer(buf, dtype=np.float32)[: n.value].copy()

    def flat_gradient(self, batch: TensorBatch) -> np.ndarray:
        """Forward+backward only; returns the flat gradient (+4 tail words) as numpy (tests)."""
        b, keep, B = self._batch_struct(batch)
        self._prepare(B)
        sc = hb.StepScalars()
        t1 = {g: max(1, v) for g, v in self._adam_t.items()}
        self._fill_scalars(sc, t1, self._current_lrs(), dp.inv_batch(B, self._dp_world))
        flat = self._dp_flat()
        hb.check(hb.lib().iqlhip_forward_backward(self._ctx, C.byref(b), C.byref(sc), flat.data_ptr(), self._stream()))
        torch.cuda.synchronize(self._dev)
        return flat.cpu().numpy()
