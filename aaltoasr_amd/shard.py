"""shard.py -- multi-GPU plumbing: one process per GPU, utterances sharded with
the reference's own rule, no collective on the scoring path.

The reference scales by running N `phone_probs -B N -I k` processes over
contiguous recipe slices (aku/Recipe.cc:63-112, aku/phone_probs.cc:134-139);
rank r of a torch.distributed job takes slice k = r+1.  The only collectives
are (a) an optional one-shot broadcast of the model parameters from rank 0
(RCCL over xGMI on GPUs, gloo in the CPU tests) and (b) the reduction of the
run statistics at the end.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np


def rank_slice(n_lines: int, world: int, rank: int) -> Tuple[int, int]:
    """(first_line, num_lines) of rank `rank` (0-based) among `world` ranks."""
    from . import capi
    if world <= 1:
        return 0, n_lines
    return capi.recipe_batch_range(n_lines, world, rank + 1)


def broadcast_model(arrays: Dict[str, np.ndarray], src: int = 0, device=None) -> Dict[str, np.ndarray]:
    """Broadcasts the model arrays (mean, var, mix_off, mix_idx, mix_w) from
    rank `src`.  Non-source ranks pass arrays of the right shape/dtype (or
    None values with a 'shapes' entry already agreed on); returns numpy arrays.
    """
    import torch
    import torch.distributed as dist
    out = {}
    # shapes first (int64 vector per array) so receivers can allocate
    names = sorted(arrays.keys())
    for name in names:
        a = arrays[name]
        meta = torch.zeros(4, dtype=torch.int64)
        if dist.get_rank() == src:
            a = np.ascontiguousarray(a)
            meta[0] = a.ndim
            for i, s in enumerate(a.shape):
                meta[1 + i] = s
            meta[3] = {np.dtype(np.float64): 0, np.dtype(np.int32): 1, np.dtype(np.float32): 2}[a.dtype]
        meta_d = meta.to(device) if device is not None else meta
        dist.broadcast(meta_d, src)
        meta = meta_d.cpu()
        shape = tuple(int(meta[1 + i]) for i in range(int(meta[0])))
        dtype = [torch.float64, torch.int32, torch.float32][int(meta[3])]
        if dist.get_rank() == src:
            t = torch.from_numpy(np.ascontiguousarray(arrays[name]))
        else:
            t = torch.empty(shape, dtype=dtype)
        t_d = t.to(device) if device is not None else t
        dist.broadcast(t_d, src)
        out[name] = t_d.cpu().numpy()
    return out


def aggregate(frames_local: int, seconds_local: float, device=None) -> Tuple[int, float]:
    """(total frames over all ranks, max seconds over ranks)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return frames_local, seconds_local
    f = torch.tensor([frames_local], dtype=torch.int64)
    s = torch.tensor([seconds_local], dtype=torch.float64)
    if device is not None:
        f, s = f.to(device), s.to(device)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    dist.all_reduce(s, op=dist.ReduceOp.MAX)
    return int(f.item()), float(s.item())
