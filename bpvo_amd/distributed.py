"""Multi-GPU sharding of batches of independent frame pairs (BASELINE.json config 5).

The hot path partitions by frame pair: every pair has its own template, pose and 6x6 system, so ranks never
exchange anything on the data path.  The only collective is ONE gather of the fixed-size result records
(32 floats per pair: pose 3x4, per-level iteration counts and statuses) — RCCL over xGMI when the process group is
"nccl" on ROCm, gloo in the CPU tests.
"""
from __future__ import annotations

RECORD_FLOATS = 32


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous block of pair indices owned by `rank` (pair i -> rank i // ceil(n/world))."""
    per = (n_total + world - 1) // world
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi


def gather_records(local_records, dst: int = 0, group=None):
    """One gather of the per-pair result records to rank `dst`.

    local_records: torch tensor [n_local, 32] on this rank's device (cuda under nccl/RCCL, cpu under gloo); all ranks
    must pass the same n_local (weak scaling: fixed pairs per GPU).  Returns the [world*n_local, 32] tensor on rank
    dst, None elsewhere."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_records
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if rank == dst:
        out = [torch.empty_like(local_records) for _ in range(world)]
        dist.gather(local_records, gather_list=out, dst=dst, group=group)
        return torch.cat(out, dim=0)
    dist.gather(local_records, gather_list=None, dst=dst, group=group)
    return None


def records_to_poses(records):
    """[n, 32] records -> ([n, 4, 4] poses, [n, 8] iterations per level, [n, 8] status per level) as numpy arrays."""
    import numpy as np

    r = records.detach().cpu().numpy() if hasattr(records, "detach") else np.asarray(records)
    n = r.shape[0]
    poses = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    poses[:, :3, :] = r[:, :12].reshape(n, 3, 4)
    return poses, r[:, 12:20].astype(np.int32), r[:, 20:28].astype(np.int32)
