"""Minimal evaluation harness for the hot path (SURVEY 8(f) "next", rank 1): load a checkpoint written by the reference's
``Agent.save_ckpt`` (agent.py:111-153: ``torch.save({"clock", "flow_state_dict", "optimizer_flow_state_dict", ...})``), read a
``raw`` dataset file (dataset/dataset_raw.py:13: ``{data_dir}/raw/{category}_{phase}.npy``, float32 ``[M,3,3]``) and reproduce the
statistic ``eval_uncondition.py:31-45`` prints: the mean over the whole test set of the per-sample log-likelihood
``ldj + base_log_prob`` (agent.py:226-229).  Everything numerical runs through the HIP path (Flow.log_prob).

    python -m rotationnormflow_amd.harness --ckpt exps/.../ckpt_iteration50000.pth --data data/raw/peak_test.npy \\
           [--config settings/raw.yml] [--batch-size 1048576]
"""
from __future__ import annotations

import argparse
import contextlib
import io

import numpy as np
import torch

from .configs import load_yaml_config, make_config
from .flow.flow import Flow


def load_reference_checkpoint(path, map_location="cpu") -> dict:
    """-> the flow's state dict from a reference checkpoint (or from a bare state-dict file).  Strips a DataParallel
    ``module.`` prefix if present (agent.py:133 saves ``flow.module.state_dict()``, older dumps may not)."""
    obj = torch.load(path, map_location=map_location, weights_only=False)
    sd = obj.get("flow_state_dict", obj) if isinstance(obj, dict) else obj
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


def build_flow_from_checkpoint(config, ckpt_path, device="cuda") -> Flow:
    with contextlib.redirect_stdout(io.StringIO()):
        flow = Flow(config)
    missing = flow.load_state_dict(load_reference_checkpoint(ckpt_path), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return flow.to(device).eval()


def load_raw_rotations(path) -> torch.Tensor:
    data = np.load(path)
    if data.ndim != 3 or data.shape[1:] != (3, 3):
        raise ValueError(f"{path}: expected a [M,3,3] array of rotation matrices, got {data.shape}")
    return torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32))


def mean_log_likelihood(flow: Flow, rotations: torch.Tensor, base=None, batch_size: int = 1 << 20, device="cuda") -> float:
    """eval_uncondition.py:31-45: np.mean over the test set of (ldj + base log-prob).  fp64 accumulation on the device."""
    total = torch.zeros(2, dtype=torch.float64, device=device)
    with torch.no_grad():
        for lo in range(0, rotations.shape[0], batch_size):
            total += flow.log_prob(rotations[lo:lo + batch_size].to(device), base=base)["sum"]
    return float(total[0] / total[1])


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ckpt", required=True)
    ap.add_argument("--data", required=True)
    ap.add_argument("--config", action="append", default=[], help="settings/*.yml file(s); later files override earlier ones")
    ap.add_argument("--layers", type=int)
    ap.add_argument("--segments", type=int)
    ap.add_argument("--rot")
    ap.add_argument("--batch-size", type=int, default=1 << 20)
    args = ap.parse_args(argv)
    over = {k: v for k, v in (("layers", args.layers), ("segments", args.segments), ("rot", args.rot)) if v is not None}
    config = load_yaml_config(*args.config, **over) if args.config else make_config(**over)
    flow = build_flow_from_checkpoint(config, args.ckpt)
    print(mean_log_likelihood(flow, load_raw_rotations(args.data), batch_size=args.batch_size))


if __name__ == "__main__":
    main()


def estimate_rotations(flow: Flow, feature: torch.Tensor, queries: torch.Tensor = None, base=None, number_queries: int = 500):
    """Pose estimate per feature row, as ``Agent.eval_acc`` does (agent.py:238-283): push ``number_queries`` base samples per row
    through ``Flow.inverse`` and keep the sample with the largest ``-ldj + base_log_prob``.

    feature [B,F] (cuda); queries [Q,3,3] shared by all rows (the reference's ``sd.generate_queries``), or ``base`` = a
    ``MatrixFisherN`` with B rows to draw them from (``pretrain_fisher``).  Returns (est_rotation [B,3,3], log_prob [B,Q])."""
    B = feature.shape[0]
    with torch.no_grad():
        if base is not None:
            sample = base._sample(number_queries).reshape(-1, 3, 3)                     # agent.py:248-251
            base_ll = base._log_prob(sample).reshape(B, -1)
            Q = number_queries
        else:
            Q = queries.shape[0]
            sample = queries[None].expand(B, Q, 3, 3).reshape(-1, 3, 3).contiguous()    # agent.py:253-258
            base_ll = torch.zeros(B, Q, device=feature.device)
        feat = feature[:, None, :].expand(B, Q, feature.shape[1]).reshape(B * Q, -1).contiguous()   # agent.py:240-244
        samples, ldj = flow.inverse(sample, feat)
        log_prob = -ldj.reshape(B, Q) + base_ll                                         # agent.py:262-263
        best = torch.argmax(log_prob, dim=-1)
        est = samples.reshape(B, Q, 3, 3)[torch.arange(B, device=best.device), best]
    return est, log_prob
