#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (/root/reference) in this container.

Run from anywhere:   python tests/golden/make_golden.py [case ...]

* Imports the reference's ``flow.flow.Flow`` and ``utils.fisher.MatrixFisherN`` with the two third-party stand-ins
  under oracle/stubs/ (pytorch3d.transforms, nflows.distributions) ahead of /root/reference on sys.path.
* Fills the reference's own ``state_dict`` by the recipe in rotationnormflow_amd/synth.py (sorted-key order), feeds
  recipe inputs, and stores ONLY outputs (fp32 run and fp64 run) + checksums of the regenerated inputs/weights.
* Nothing of the reference travels: fixtures are arrays of numbers.  The GPU box never runs this script.
"""
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

if not os.path.isdir(REF):
    sys.exit("reference tree not present; golden fixtures can only be regenerated in the build container")

sys.path[:0] = [os.path.join(REPO, "oracle", "stubs"), REF, REPO]

import numpy as np  # noqa: E402
import torch  # noqa: E402

import flow.flow as ref_flow_mod  # noqa: E402  (the reference's package)
import utils.fisher as ref_fisher_mod  # noqa: E402

assert ref_flow_mod.__file__.startswith(REF), ref_flow_mod.__file__
assert ref_fisher_mod.__file__.startswith(REF), ref_fisher_mod.__file__

from rotationnormflow_amd import synth  # noqa: E402
from rotationnormflow_amd.configs import make_config  # noqa: E402
from tests.golden.cases import CASES  # noqa: E402


def crc(a: np.ndarray) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def build_reference_flow(cfg, dtype):
    torch.set_default_dtype(dtype)          # mobiusflow.py:80 uses torch.empty(size) in the DEFAULT dtype
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):   # flow.py:50 prints the layer count
        fl = ref_flow_mod.Flow(cfg)
    return fl.to(dtype).eval()


def run_case(name, spec):
    cfg = make_config(**spec["cfg"])
    n = spec["n"]
    rot_in = synth.uniform_rotations(n, seed=spec["rseed"])
    fdim = 0
    out = {}
    res = {}
    for tag, dtype in (("32", torch.float32), ("64", torch.float64)):
        fl = build_reference_flow(cfg, dtype)
        sd = fl.state_dict()
        shapes = {k: tuple(v.shape) for k, v in sd.items()}
        weights = synth.fill_state_dict(shapes, seed=spec["wseed"], regime=spec["regime"])
        fl.load_state_dict({k: torch.from_numpy(v).to(dtype) for k, v in weights.items()})
        fdim = fl.feature_dim
        feat = None
        if cfg.condition:
            feat = torch.from_numpy(synth.features(n, fdim, seed=spec["rseed"] + 1000) * np.float32(synth.feature_scale(spec["regime"]))).to(dtype)
        R = torch.from_numpy(rot_in).to(dtype)
        with torch.no_grad():
            if spec["direction"] == "forward":
                Rt, ldj = fl(R, feat)
            else:
                Rt, ldj = fl.inverse(R, feat)
            res[tag] = (Rt, ldj)
            out["rot" + tag] = Rt.numpy().copy()
            out["ldj" + tag] = ldj.numpy().copy()
            if spec["fisher"] is not None:
                A = torch.from_numpy(synth.fisher_A(spec["fisher"])).to(dtype)
                dist = ref_fisher_mod.MatrixFisherN(A)
                # forward: base density of the flow output; inverse: base density of the given base samples
                arg = Rt if spec["direction"] == "forward" else R
                out["fisher" + tag] = dist._log_prob(arg).numpy().copy()
        if tag == "32":
            out["n_layers"] = np.int64(len(fl.layers))
            out["n_keys"] = np.int64(len(sd))
            out["keys"] = np.array(sorted(sd.keys()))
            out["w_crc"] = np.int64(crc(np.concatenate([weights[k].ravel() for k in sorted(weights)])))
    torch.set_default_dtype(torch.float32)
    out["in_crc"] = np.int64(crc(rot_in))
    out["feature_dim"] = np.int64(fdim)
    d = np.abs(out["ldj32"].astype(np.float64) - out["ldj64"])
    print(f"{name:18s} layers={int(out['n_layers']):3d} keys={int(out['n_keys']):4d} n={n:5d} "
          f"ldj64 range [{out['ldj64'].min():8.3f},{out['ldj64'].max():8.3f}] "
          f"|ldj32-ldj64| mean {d.mean():.2e} p99 {np.quantile(d, .99):.2e} max {d.max():.2e}")
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def grad_case_loss_weights(n, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal(n), rng.standard_normal((n, 3, 3))


def run_inverse_grad_case(name, spec):
    """Gradients of loss = sum(a * ldj) + sum(B * R_out) through the reference's Flow.inverse (BinFind.backward) or, with
    direction="forward", Flow.forward; fp64."""
    cfg = make_config(**spec["cfg"])
    n = spec["n"]
    dtype = torch.float64
    fl = build_reference_flow(cfg, dtype).train()
    sd = fl.state_dict()
    weights = synth.fill_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed=spec["wseed"], regime=spec["regime"])
    fl.load_state_dict({k: torch.from_numpy(v).to(dtype) for k, v in weights.items()})
    R = torch.from_numpy(synth.uniform_rotations(n, seed=spec["rseed"])).to(dtype).requires_grad_(True)
    feat = None
    if cfg.condition:
        feat = torch.from_numpy(synth.features(n, fl.feature_dim, seed=spec["rseed"] + 1000)).to(dtype).requires_grad_(True)
    a, B = grad_case_loss_weights(n, spec["rseed"] + 7)
    # direction "forward": what agent.py:75-92 differentiates (Flow.forward); default: Flow.inverse
    Rt, ldj = fl(R, feat) if spec.get("direction") == "forward" else fl.inverse(R, feat)
    loss = (torch.from_numpy(a) * ldj).sum() + (torch.from_numpy(B) * Rt).sum()
    loss.backward()
    out = {"loss": np.float64(loss.item()), "g_rot": R.grad.numpy().copy(), "rot_out": Rt.detach().numpy().copy(), "ldj": ldj.detach().numpy().copy()}
    if feat is not None:
        out["g_feat"] = feat.grad.numpy().copy()
    for k, prm in fl.named_parameters():
        out["g:" + k] = prm.grad.numpy().copy() if prm.grad is not None else np.zeros(tuple(prm.shape))
    torch.set_default_dtype(torch.float32)
    gmax = max(float(np.abs(v).max()) for k, v in out.items() if k.startswith("g:"))
    print(f"{name:22s} n={n} params={sum(1 for k in out if k.startswith('g:'))} loss={out['loss']:.6f} max|grad|={gmax:.3e}")
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def main():
    from tests.golden.cases import GRAD_CASES
    torch.manual_seed(0)
    names = [n for n in sys.argv[1:] if n in CASES] or ([] if len(sys.argv) > 1 else list(CASES))
    for name in names:
        run_case(name, CASES[name])
    gnames = [n for n in sys.argv[1:] if n in GRAD_CASES] or ([] if len(sys.argv) > 1 else list(GRAD_CASES))
    for name in gnames:
        run_inverse_grad_case(name, GRAD_CASES[name])


if __name__ == "__main__":
    main()


def run_fisher_sampler_case():
    """MatrixFisherN._sample with a fixed torch seed: pins the oracle's restatement of the rejection sampler (same RNG
    consumption order) sample by sample."""
    A = torch.from_numpy(np.concatenate([synth.fisher_A("diag531"), synth.fisher_A("tilted")], axis=0))
    torch.manual_seed(1234)
    dist = ref_fisher_mod.MatrixFisherN(A.clone())
    samples = dist._sample(256)
    lp = dist._log_prob(samples.reshape(-1, 3, 3))
    np.savez_compressed(os.path.join(HERE, "fisher_sampler.npz"), A=A.numpy(), samples=samples.numpy(), log_prob=lp.numpy())
    print("fisher_sampler: ", tuple(samples.shape), "mean log_prob", float(lp.mean()))


if __name__ == "__main__" and (len(sys.argv) == 1 or "fisher_sampler" in sys.argv):
    run_fisher_sampler_case()


def run_fisher_grad_case():
    """The reference's own autograd through MatrixFisherN(A, norm_type)._log_prob w.r.t. A (agent.py:57-65 keeps a predicted A in the
    graph) and w.r.t. the rotations, for the two normaliser approximations that are closed forms (norm_type 0 and 1), fp32 and fp64.
    Six matrices (one with det < 0, one small) x 5 rotations each; the loss is a fixed random weighting of the log-densities."""
    rng = np.random.RandomState(77)
    A = np.concatenate([synth.fisher_A("diag531"), synth.fisher_A("tilted"), rng.randn(4, 3, 3)], axis=0)
    A[3] *= 0.2
    if np.linalg.det(A[4]) > 0:
        A[4, :, 0] *= -1.0
    R = synth.uniform_rotations(A.shape[0] * 5, seed=78).astype(np.float64)
    g = rng.randn(R.shape[0])
    out = dict(A=A, R=R, g=g)
    for nt in (0, 1):
        for dtype, tag in ((torch.float64, "64"), (torch.float32, "32")):
            At = torch.from_numpy(A).to(dtype).requires_grad_(True)
            Rt = torch.from_numpy(R).to(dtype).requires_grad_(True)
            dist = ref_fisher_mod.MatrixFisherN(At, norm_type=nt)
            lp = dist._log_prob(Rt)
            (lp * torch.from_numpy(g).to(dtype)).sum().backward()
            out[f"logp_t{nt}_{tag}"] = lp.detach().numpy().copy()
            out[f"gA_t{nt}_{tag}"] = At.grad.numpy().copy()
            out[f"gR_t{nt}_{tag}"] = Rt.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "fisher_grad.npz"), **out)
    print("fisher_grad: ", {k: tuple(v.shape) for k, v in out.items() if k.startswith("gA")})


if __name__ == "__main__" and (len(sys.argv) == 1 or "fisher_grad" in sys.argv):
    run_fisher_grad_case()
