"""Minimal evaluation harness for the hot path (SURVEY 8(f) "next", rank 1): load a checkpoint written by the reference's
``Agent.save_ckpt`` (agent.py:111-153: ``torch.save({"clock", "flow_state_dict", "optimizer_flow_state_dict", ...})``), read a
``raw`` dataset file (dataset/dataset_raw.py:13: ``{data_dir}/raw/{category}_{phase}.npy``, float32 ``[M,3,3]``) and reproduce the
statistic ``eval_uncondition.py:31-45`` prints: the mean over the whole test set of the per-sample log-likelihood
``ldj + base_log_prob`` (agent.py:226-229).  Everything numerical runs through the HIP path (Flow.log_prob).

    python -m rotationnormflow_amd.harness --ckpt exps/.../ckpt_iteration50000.pth --data data/raw/peak_test.npy \\
           [--config settings/raw.yml] [--batch-size 1048576]

``train_uncondition`` is the matching minimal training loop (train_uncondition.py:37-90 + agent.py:75-92 for the unconditional
``raw`` recipe): Adam on ``mean(-ldj)`` over shuffled mini-batches, periodic test log-likelihood, checkpoints in the reference's
format so that ``Agent.load_ckpt`` (agent.py:171-198) and this module read them back.

    python -m rotationnormflow_amd.harness --train data/raw/peak_train.npy --data data/raw/peak_test.npy --ckpt out.pth \\
           [--config settings/raw.yml] [--iterations 50000] [--train-batch 1024] [--lr 1e-4]
"""
from __future__ import annotations

import argparse
import contextlib
import io
import os

import numpy as np
import torch

from .configs import load_yaml_config, make_config
from . import runtime
from .flow.flow import Flow


def load_reference_checkpoint(path, map_location="cpu") -> dict:
    """-> the flow's state dict from a reference checkpoint (or from a bare state-dict file).  Strips a DataParallel
    ``module.`` prefix if present (agent.py:133 saves ``flow.module.state_dict()``, older dumps may not)."""
    obj = torch.load(path, map_location=map_location, weights_only=False)
    sd = obj.get("flow_state_dict", obj) if isinstance(obj, dict) else obj
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


FEATURE_SCALE_SUFFIX = ".rnf.json"


def read_sidecar(ckpt_path) -> dict:
    """``<ckpt>.rnf.json`` as a dictionary ({} when there is none): "feature_mean_square" (the equalisation's calibration input) and, round 6,
    "rootfinder_first_order" (3 | 4, ``Flow.set_rootfinder_order``).  The reference's checkpoint layout (agent.py:132-146) has no place for
    either and must stay loadable by the reference, hence a file NEXT to the checkpoint."""
    import json
    side = str(ckpt_path) + FEATURE_SCALE_SUFFIX
    if not os.path.exists(side):
        return {}
    with open(side) as fh:
        return dict(json.load(fh))


def read_feature_scale(ckpt_path):
    """The calibration a checkpoint's sidecar carries (``<ckpt>.rnf.json``: {"feature_mean_square": m_f}), or None."""
    value = read_sidecar(ckpt_path).get("feature_mean_square")
    return None if value is None else float(value)


def write_rootfinder_order(ckpt_path, order: int) -> None:
    """Record the root finder's first-pass order (3 | 4) in the checkpoint's sidecar, beside whatever it already holds."""
    import json
    if order not in (3, 4):
        raise ValueError(f"root-finder first-pass order must be 3 or 4, got {order!r}")
    side = read_sidecar(ckpt_path)
    side["rootfinder_first_order"] = int(order)
    with open(str(ckpt_path) + FEATURE_SCALE_SUFFIX, "w") as fh:
        json.dump(side, fh)


def write_feature_scale(ckpt_path, flow_or_value, features=None) -> float:
    """Write the sidecar: the flow's fixed calibration (``Flow.set_feature_scale`` / ``calibrate_feature_scale``), a number, or -- with
    ``features`` -- the mean square measured on that batch.  Conditional flows whose features are not of unit scale want one, so that every
    process that loads the checkpoint packs the same images without seeing data first."""
    import json
    if features is not None:
        value = runtime.feature_mean_square(features)
    elif isinstance(flow_or_value, (int, float)):
        value = runtime.quantise_feature_ms(float(flow_or_value))
    else:
        value = getattr(flow_or_value, "_feature_ms_fixed", None)
        if value is None:
            raise ValueError("the flow has no fixed feature scale: call flow.calibrate_feature_scale(features) first, or pass features=")
    side = read_sidecar(ckpt_path)
    side["feature_mean_square"] = float(value)
    with open(str(ckpt_path) + FEATURE_SCALE_SUFFIX, "w") as fh:
        json.dump(side, fh)
    return float(value)


def build_flow_from_checkpoint(config, ckpt_path, device="cuda") -> Flow:
    with contextlib.redirect_stdout(io.StringIO()):
        flow = Flow(config)
    missing = flow.load_state_dict(load_reference_checkpoint(ckpt_path), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    flow = flow.to(device).eval()
    side = read_sidecar(ckpt_path)
    ms = side.get("feature_mean_square") if getattr(config, "condition", 0) else None
    if ms is not None:
        flow.set_feature_scale(float(ms))
    if side.get("rootfinder_first_order") is not None:
        flow.set_rootfinder_order(int(side["rootfinder_first_order"]))
    return flow


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


def expand_optimizer_state(flow: Flow, state: dict) -> dict:
    """Optimizer state of a FLATTENED flow in its in-memory form (one entry) -> the layout the reference's ``optim.Adam(flow.parameters())``
    has (one entry per parameter tensor, agent.py:23,143).  ``optimizer.state_dict()`` of an optimizer built over a flattened flow already
    returns that layout (rotationnormflow_amd/flatopt.py hooks it); such a state is returned unchanged."""
    if not flow.is_flat:
        return state
    from . import flatopt
    return flatopt.expand_state([{"params": [flow._parameters["_flat"]]}], state)


def flatten_optimizer_state(flow: Flow, state: dict) -> dict:
    """The inverse: an optimizer state in the reference's per-tensor layout (a checkpoint ``Agent.save_ckpt`` wrote) -> the one-entry
    state of an optimizer over the flattened flow's parameter (what ``optimizer.load_state_dict`` does by itself through flatopt)."""
    if not flow.is_flat:
        return state
    from . import flatopt
    n = len(flow._flat_slots)
    if len(state["param_groups"]) != 1 or len(state["param_groups"][0]["params"]) not in (1, n):
        raise ValueError(f"optimizer state covers {[len(g['params']) for g in state['param_groups']]} parameters, the flow has {n} tensors")
    return flatopt.flatten_state([{"params": [flow._parameters["_flat"]]}], state)


def save_reference_checkpoint(path, flow: Flow, optimizer, epoch: int, minibatch: int, iteration: int):
    """The dictionary ``Agent.save_ckpt`` writes for an unconditional flow (agent.py:132-151; clock: utils/utils.py:36-45).  The optimizer
    state of a flattened flow is written in the reference's per-tensor layout (``optimizer.state_dict()`` returns it: flatopt)."""
    torch.save({
        "clock": {"epoch": epoch, "minibatch": minibatch, "iteration": iteration},
        "flow_state_dict": {k: v.detach().cpu() for k, v in flow.state_dict().items()},
        "optimizer_flow_state_dict": expand_optimizer_state(flow, optimizer.state_dict()),
    }, path)


def host_preprocess_layers(flow) -> list:
    """Names of the layer classes of ``flow`` whose training tensors go through host-side linear algebra (rot='16Rot' / '16UnRot',
    '9TransLSVD' / '9TransRSVD' / '9TransRSmith'): such flows cannot be captured into a HIP graph."""
    return sorted({type(l).__name__ for l in flow.layers if getattr(l, "_rnf_host_preprocess", False) or getattr(l, "_rnf_no_graph", False)})


class GraphedTrainStep:
    """One training iteration (agent.py:75-92: forward, loss, zero_grad, backward, optimizer step) captured ONCE as a HIP graph and
    replayed: the iteration is launch-bound on the host (264 parameter tensors go through autograd and the optimizer one by one), the
    graph replays the same dozen device launches without any of it.

        step = GraphedTrainStep(flow, optimizer, rotation_shape=(1024, 3, 3))          # optimizer: Adam(..., capturable=True)
        loss = step(batch)                                                             # device tensor, valid until the next call

    Static shapes: every batch must have ``rotation_shape`` (and ``feature_shape``).  ``base``: optional MatrixFisherN with a frozen A
    (its term is added to the loss as in agent.py:58-65)."""

    def __init__(self, flow: Flow, optimizer, rotation_shape, feature_shape=None, base=None, warmup: int = 3, device="cuda"):
        from . import runtime
        blockers = host_preprocess_layers(flow)
        if blockers:
            raise RuntimeError(f"GraphedTrainStep: {', '.join(blockers)} build their training tensors on the host (a 4x4 / 3x3 SVD or "
                               "Gram-Schmidt per iteration: a device->host copy, illegal during stream capture); train this flow eagerly "
                               "(train_uncondition(..., graph=False))")
        self._runtime = runtime
        self.flow, self.optimizer, self.base = flow, optimizer, base
        self.rotation = torch.eye(3, device=device).expand(*rotation_shape).contiguous()
        self.feature = torch.zeros(feature_shape, device=device) if feature_shape is not None else None
        # The warm-up iterations torch.cuda.graphs needs (lazy optimizer state, allocator pools) run on placeholder data, so the
        # parameters and the optimizer state are put back IN PLACE afterwards (the graph must keep seeing the same tensors).
        params = [p for g in optimizer.param_groups for p in g["params"]]
        saved_params = [p.detach().clone() for p in params]
        saved_state = {id(p): {k: v.clone() for k, v in optimizer.state.get(p, {}).items() if torch.is_tensor(v)} for p in params}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # off the default stream, as torch.cuda.graphs requires
            for _ in range(warmup):
                optimizer.zero_grad(set_to_none=True)
                self._loss().backward()
                optimizer.step()
            with torch.no_grad():
                for p, old in zip(params, saved_params):
                    p.copy_(old)
                    for k, v in optimizer.state.get(p, {}).items():
                        if torch.is_tensor(v):
                            before = saved_state[id(p)].get(k)
                            v.copy_(before) if before is not None else v.zero_()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = self._loss()
            self.loss.backward()
            optimizer.step()

    def _loss(self):
        rot, ldj = self.flow(self.rotation, self.feature)
        loss = (-ldj).mean()
        if self.base is not None:
            loss = loss - self.base._log_prob(rot).mean()
        return loss

    def __call__(self, rotation, feature=None):
        self.rotation.copy_(rotation)
        if self.feature is not None:
            self.feature.copy_(feature)
        self.graph.replay()
        self._runtime.note_training_step()                  # parameters changed: host-packed blobs of the eval path are stale
        return self.loss


def train_uncondition(flow: Flow, train_rotations: torch.Tensor, iterations: int, batch_size: int = 1024, lr: float = 1e-4,
                      seed: int = 42, test_rotations: torch.Tensor = None, val_every: int = 0, ckpt_path=None, save_every: int = 0,
                      base=None, device="cuda", log=print, graph: bool = True, data_parallel=None):
    """Maximum-likelihood training of an unconditional flow on a ``raw`` rotation set.  One iteration = the reference's
    ``Agent.train_func`` (agent.py:75-92): loss = mean(-ldj) (- mean base log-prob if ``base`` is given), zero_grad, backward, Adam
    step -- here three HIP launches (device packer, fused forward, fused backward) plus the optimizer; with ``graph`` (default) the
    whole iteration is captured once as a HIP graph and replayed (GraphedTrainStep), which removes the per-tensor host work.
    Returns the list of (iteration, train loss) pairs sampled every 100 iterations and the final test log-likelihood (or None)."""
    flow = flow.to(device).train()
    gen = torch.Generator().manual_seed(seed)
    data = train_rotations.to(device)
    n = data.shape[0]
    # data parallel: every rank draws the SAME global mini-batch (same seed) and trains on its own contiguous slice of it; the
    # gradient blob is averaged over the ranks by one all-reduce inside backward (dist.data_parallel_training), so each rank takes
    # the optimizer step of the global batch and the replicas stay identical.  Over RCCL the all-reduce is captured into the HIP
    # graph of the iteration like any other launch (stream-ordered); over gloo (host collectives) the iteration runs eagerly.
    import torch.distributed as tdist
    world = tdist.get_world_size() if (tdist.is_available() and tdist.is_initialized()) else 1
    rank = tdist.get_rank() if world > 1 else 0
    from .dist import collectives_are_capturable, data_parallel_training, shard_bounds
    if world > 1 or data_parallel:                       # data_parallel=True forces the gradient all-reduce on a one-rank group too
        data_parallel_training(flow, even_alone=bool(data_parallel) and world == 1)
        if graph and not collectives_are_capturable():    # RCCL collectives are stream-ordered and captured with the rest of the
            graph = False                                 # iteration; gloo's run on the host: eager iterations then
    if graph and host_preprocess_layers(flow):
        log(f"train_uncondition: {', '.join(host_preprocess_layers(flow))} need host-side linear algebra every iteration; training eagerly "
            "(no HIP graph)")
        graph = False
    opt = torch.optim.Adam(flow.parameters(), lr=lr, fused=True, capturable=graph)   # one launch instead of a dozen foreach kernels
    batch_size = min(batch_size, n)
    if graph and world > 1 and batch_size % world:        # the graph has static shapes: every rank's slice must have the same size
        raise ValueError(f"train_uncondition: batch_size {batch_size} must be a multiple of the world size {world} for graphed "
                         "data-parallel training (or pass graph=False)")
    gstep = GraphedTrainStep(flow, opt, (batch_size // world, 3, 3), base=base, device=device) if graph else None
    history, it, epoch = [], 0, 0
    while it < iterations:
        perm = torch.randperm(n, generator=gen).to(device)
        for mb, lo in enumerate(range(0, n - batch_size + 1, batch_size)):
            batch = data[perm[lo:lo + batch_size]]
            if world > 1:
                b0, b1 = shard_bounds(batch.shape[0], rank, world)
                batch = batch[b0:b1]
            if gstep is not None:
                loss = gstep(batch)
            else:
                rot, ldj = flow(batch)
                loss = (-ldj).mean()
                if base is not None:
                    loss = loss - base._log_prob(rot).mean()
                opt.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()
            it += 1
            if it % 100 == 0 or it == iterations:
                history.append((it, float(loss.detach())))
            if val_every and test_rotations is not None and it % val_every == 0:
                log(f"iteration {it}: train loss {float(loss.detach()):.6f}  test log-likelihood "
                    f"{mean_log_likelihood(flow, test_rotations, base=base, device=device):.6f}")
                flow.train()
            if ckpt_path and save_every and it % save_every == 0:
                save_reference_checkpoint(ckpt_path, flow, opt, epoch, mb + 1, it)
            if it >= iterations:
                break
        epoch += 1
    if ckpt_path:
        save_reference_checkpoint(ckpt_path, flow, opt, epoch, 0, it)
    final = mean_log_likelihood(flow, test_rotations, base=base, device=device) if test_rotations is not None else None
    return history, final


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ckpt", required=True, help="checkpoint to evaluate, or (with --train) to write")
    ap.add_argument("--data", required=True, help="raw test set, [M,3,3] float32 .npy")
    ap.add_argument("--train", help="raw training set: train an unconditional flow first, then evaluate it on --data")
    ap.add_argument("--iterations", type=int, default=50000)
    ap.add_argument("--train-batch", type=int, default=1024)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--config", action="append", default=[], help="settings/*.yml file(s); later files override earlier ones")
    ap.add_argument("--layers", type=int)
    ap.add_argument("--segments", type=int)
    ap.add_argument("--rot")
    ap.add_argument("--batch-size", type=int, default=1 << 20)
    args = ap.parse_args(argv)
    over = {k: v for k, v in (("layers", args.layers), ("segments", args.segments), ("rot", args.rot)) if v is not None}
    config = load_yaml_config(*args.config, **over) if args.config else make_config(**over)
    if args.train:
        torch.manual_seed(args.seed)
        with contextlib.redirect_stdout(io.StringIO()):
            flow = Flow(config)
        _, final = train_uncondition(flow, load_raw_rotations(args.train), args.iterations, args.train_batch, args.lr, args.seed,
                                     test_rotations=load_raw_rotations(args.data), val_every=1000, ckpt_path=args.ckpt, save_every=5000)
        print(final)
        return
    flow = build_flow_from_checkpoint(config, args.ckpt)
    print(mean_log_likelihood(flow, load_raw_rotations(args.data), batch_size=args.batch_size))


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
        # agent.py:240-244 repeats every feature row Q times; here the rows are shared inside the kernels (one projection per image)
        samples, ldj = flow.inverse(sample, feature, feature_repeat=Q)
        log_prob = -ldj.reshape(B, Q) + base_ll                                         # agent.py:262-263
        best = torch.argmax(log_prob, dim=-1)
        est = samples.reshape(B, Q, 3, 3)[torch.arange(B, device=best.device), best]
    return est, log_prob


def matrix_to_quaternion(R: torch.Tensor) -> torch.Tensor:
    """[B,3,3] rotations -> unit quaternions [B,4] (real part first), largest-component branch per row."""
    m = R.reshape(-1, 3, 3)
    m00, m11, m22 = m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]
    cand = torch.stack([
        torch.stack([1 + m00 + m11 + m22, m[:, 2, 1] - m[:, 1, 2], m[:, 0, 2] - m[:, 2, 0], m[:, 1, 0] - m[:, 0, 1]], -1),
        torch.stack([m[:, 2, 1] - m[:, 1, 2], 1 + m00 - m11 - m22, m[:, 0, 1] + m[:, 1, 0], m[:, 0, 2] + m[:, 2, 0]], -1),
        torch.stack([m[:, 0, 2] - m[:, 2, 0], m[:, 0, 1] + m[:, 1, 0], 1 - m00 + m11 - m22, m[:, 1, 2] + m[:, 2, 1]], -1),
        torch.stack([m[:, 1, 0] - m[:, 0, 1], m[:, 0, 2] + m[:, 2, 0], m[:, 1, 2] + m[:, 2, 1], 1 - m00 - m11 + m22], -1)], 1)
    best = torch.argmax(torch.stack([cand[:, i, i] for i in range(4)], -1), -1)
    q = cand[torch.arange(m.shape[0], device=m.device), best]
    return q / q.norm(dim=-1, keepdim=True)


def refine_rotations(flow: Flow, feature, rotations: torch.Tensor, steps: int = 100, lr: float = 1e-4, base=None):
    """Pose refinement of ``eval.py``'s ``nll_grad`` mode (eval.py:464-480): gradient ascent of the log-density over the query
    rotation, parameterised by a quaternion; every step is the FIRST step of a fresh Adam (eval.py:470-471 re-creates the optimizer
    inside the loop), i.e. q <- q - lr * g / (|g| + 1e-8), followed by renormalisation.  The gradient w.r.t. the rotation comes from
    the backward sweep with the parameter gradients switched off (the flow's parameters are frozen for the duration).
    rotations [B,3,3], feature [B,F] or None.  Returns the refined rotations [B,3,3]."""
    from .utils.fisher import quaternion_to_matrix
    flags = [p.requires_grad for p in flow.parameters()]
    for p in flow.parameters():
        p.requires_grad_(False)
    try:
        q = matrix_to_quaternion(rotations.detach()).to(torch.float32)
        for _ in range(steps):
            q = q.detach().requires_grad_(True)
            with torch.enable_grad():
                rot, ldj = flow(quaternion_to_matrix(q), feature)
                loss = -ldj.mean()
                if base is not None:
                    loss = loss - base._log_prob(rot).mean()
                (g,) = torch.autograd.grad(loss, q)
            q = q.detach() - lr * g / (g.abs() + 1e-8)
            q = q / q.norm(dim=-1, keepdim=True)
        return quaternion_to_matrix(q.detach())
    finally:
        for p, f in zip(flow.parameters(), flags):
            p.requires_grad_(f)


def min_geodesic_distance(est_rotation: torch.Tensor, gt_rotation: torch.Tensor) -> torch.Tensor:
    """utils/utils.py:231-235: angle (radians) between each estimate [B,3,3] and the closest of its ground truths [B,K,3,3] (or [B,3,3])."""
    from . import _lib
    est = est_rotation.reshape(-1, 3, 3).to(torch.float32).contiguous()
    n = est.shape[0]
    gt = gt_rotation.reshape(n, -1, 3, 3).to(device=est.device, dtype=torch.float32).contiguous()
    if not est.is_cuda:
        raise RuntimeError("rotationnormflow_amd runs on the GPU only (no CPU fallback)")
    out = torch.empty(n, dtype=torch.float32, device=est.device)
    with torch.cuda.device(est.device):
        _lib.check(_lib.lib().rnf_min_geodesic(est.data_ptr(), gt.data_ptr(), n, gt.shape[1], out.data_ptr(),
                                               torch.cuda.current_stream(est.device).cuda_stream))
    return out


def pose_accuracy(flow: Flow, feature, gt_rotation, queries=None, base=None, number_queries: int = 500, thresholds_deg=(15.0, 30.0)):
    """What ``Agent.eval_acc`` + ``eval.py`` report per batch (agent.py:238-283, utils/utils.py:208-209): arg-max pose estimate, geodesic
    error in degrees against the (possibly several) ground truths, accuracy at the thresholds.  -> dict(err_deg, est_rotation, acc)"""
    est, _ = estimate_rotations(flow, feature, queries=queries, base=base, number_queries=number_queries)
    err_deg = torch.rad2deg(min_geodesic_distance(est, gt_rotation))
    return dict(err_deg=err_deg, est_rotation=est, acc={t: float((err_deg <= t).float().mean()) for t in thresholds_deg})


if __name__ == "__main__":
    main()
