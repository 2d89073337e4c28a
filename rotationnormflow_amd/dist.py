"""Multi-GPU harness: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), batch axis sharded,
no data-path collective; the only exchange is ONE all-reduce of {sum_i log p_i, count} (16 bytes, fp64) per evaluation.

Replaces the reference's single-process nn.DataParallel (agent.py:22,40), which re-broadcasts ~2.8 MB of parameters,
scatters the batch and gathers the outputs on device 0 on every forward.
"""
from __future__ import annotations

import torch


def shard_bounds(n: int, rank: int, world: int):
    """Contiguous split of the batch axis, same partition as torch's scatter/chunk on dim 0 (ceil-sized chunks)."""
    per = -(-n // world)
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


def all_reduce_nll(sum_count: torch.Tensor) -> torch.Tensor:
    """sum_count: float64 [2] = {sum of per-sample log-probs, sample count} of this rank's shard (device tensor).
    Returns the all-reduced {global sum, global count} (in place).  No-op without an initialised process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(sum_count, op=dist.ReduceOp.SUM)
    return sum_count


def sharded_mean_nll(evaluate, rotation, feature=None, rank: int = 0, world: int = 1):
    """evaluate(rotation_shard, feature_shard) -> float64 [2] {sum log p, count} on the shard's device.
    Every rank passes the SAME global batch (or only its own rows are touched); returns (mean NLL, {sum, count})."""
    lo, hi = shard_bounds(rotation.shape[0], rank, world)
    part = evaluate(rotation[lo:hi], None if feature is None else feature[lo:hi])
    tot = all_reduce_nll(part.clone())
    return -(tot[0] / tot[1]), tot


def flow_evaluator(flow, base=None):
    """evaluate() for sharded_mean_nll built on the fused HIP density evaluation (Flow.log_prob)."""
    def evaluate(rot, feat):
        if rot.shape[0] == 0:
            return torch.zeros(2, dtype=torch.float64, device=rot.device)
        return flow.log_prob(rot, feat, base=base)["sum"]
    return evaluate
