"""Dataset ingest on the device: the reduction behind `compute_mean_std`
(algorithms/finetune/iql.py:77-80) and the in-place `normalize_states` (:83-84) of a packed replay buffer, in
libiqlhip.so.  At 10 M rows the numpy forms cost seconds (two float32 passes plus temporaries of the dataset's size);
here the rows are uploaded once, reduced where they lie and normalised in place.

The reference's call order (finetune/iql.py:628-640: mean/std of the dataset -> normalise both state arrays -> load)
maps to:   buf.load_d4rl_dataset(raw);  mean, std = buf.state_mean_std(eps);  buf.normalize_states_(mean, std)
Numerics: the normalisation is bit-identical to numpy's given the same mean / std; the mean / std themselves are
float64-accumulated (deterministic, fixed order): within 1e-6 relative of a float64 evaluation, whereas numpy's own
float32 axis-0 reduction (row after row) drifts by ~2e-4 at 1 M rows — tolerances stated in tests/test_hip_ingest.py.  There is no CPU fallback here: the numpy functions of the reference's
surface (`compute_mean_std`, `normalize_states`) stay what they are for host arrays.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

import iqlhip_binding as hb


def _stream(dev: torch.device) -> int:
    return torch._C._cuda_getCurrentRawStream(dev.index)


def cols_mean_std(x: torch.Tensor, eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """mean(0), std(0) + eps of a 2-D float32 device tensor whose rows may be strided (a column block of packed rows)."""
    if x.device.type != "cuda":
        raise RuntimeError("iqlhip: cols_mean_std needs a GPU tensor (numpy arrays: iql.compute_mean_std)")
    if x.dim() != 2 or x.dtype != torch.float32 or (x.shape[1] > 1 and x.stride(1) != 1):
        raise ValueError("expected a float32 [n, cols] tensor with unit column stride")
    n, c = x.shape
    if n < 1:
        raise ValueError("no rows")
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    std = torch.empty(c, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        hb.check(hb.lib().iqlhip_cols_mean_std(x.data_ptr(), x.stride(0), c, n, float(eps), mean.data_ptr(),
                                               std.data_ptr(), _stream(x.device)))
    return mean, std


def compute_mean_std_device(states: np.ndarray, eps: float, device: str = "cuda") -> Tuple[np.ndarray, np.ndarray]:
    """compute_mean_std for a host array through the device reduction (chunked upload, float64 accumulation)."""
    x = torch.as_tensor(np.ascontiguousarray(states, dtype=np.float32)).to(device, non_blocking=False)
    mean, std = cols_mean_std(x, eps)
    return mean.cpu().numpy(), std.cpu().numpy()
