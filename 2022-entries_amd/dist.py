"""Multi-GPU MSM: disjoint base/scalar slices per rank, one tiny exchange, host fold.

An MSM is a sum over pairs, so rank g of G owns the contiguous slice ``[g*ceil(N/G), ...)`` of bases (uploaded once)
and of every scalar batch, runs the full single-GPU pipeline on it and contributes ONE 144-byte partial point.
The only collective on the data path is an all-gather of G x 144 B (RCCL over xGMI when the backend is "nccl";
elliptic-curve addition is not a reduction operator RCCL knows, hence gather-then-fold).
Every rank then folds the G partials with the C ABI's ``mi355_msm_fold`` ("the final 8-point curve add").
The reference has no multi-GPU path at all (every entry hard-codes device 0: SPK msm/pippenger.cuh:400-416).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

from .msm import fold_partials, projective_bytes


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice of ``range(n)`` owned by ``rank``; slices are disjoint and cover everything."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def all_gather_partials(partial: bytes, device=None, group=None) -> List[bytes]:
    """All-gather one projective image per rank (144 B for G1, 288 B for G2; uint8 tensor, on ``device`` for nccl)."""
    import torch
    import torch.distributed as dist

    nb = len(partial)
    if nb not in (144, 288):
        raise ValueError("partial must be a 144-byte (G1) or 288-byte (G2) projective image")
    world = dist.get_world_size(group)
    t = torch.frombuffer(bytearray(partial), dtype=torch.uint8)
    if device is not None:
        t = t.to(device)
    out = torch.empty(world * nb, dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    raw = out.cpu().numpy().tobytes()
    return [raw[i * nb:(i + 1) * nb] for i in range(world)]


def sharded_msm(local_msm: Callable[[], bytes], curve="bls12_377_g1", device=None, group=None) -> bytes:
    """Run ``local_msm()`` (this rank's slice -> 144-byte partial), exchange, fold.  Same bytes on every rank."""
    import torch.distributed as dist

    partial = local_msm()
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return fold_partials([partial], curve)
    return fold_partials(all_gather_partials(partial, device=device, group=group), curve)
