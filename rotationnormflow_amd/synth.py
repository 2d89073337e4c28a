"""Seeded synthetic inputs and weights-by-recipe (numpy only; no GPU, no reference import).

Shared by bench.py, the parity tests, and tests/golden/make_golden.py (which applies the same recipe to the
real reference's ``state_dict`` so that golden fixtures only need to hold *outputs*).

* rotations: Haar-uniform on SO(3) as a normalised N(0, I4) quaternion mapped to a matrix -- the semantics of
  ``pytorch3d.transforms.random_rotations`` that the reference's ``utils/sd.py:22-23`` (mode='random') uses.
* features : N(0, 1) fp32 ``[N, F]`` ("precomputed features, no ResNet", BASELINE.json configs[3]).
* weights  : every state-dict tensor filled in *sorted-key order* from one ``default_rng(seed)`` stream.
"""
from __future__ import annotations

import numpy as np

RD_SEED = 42  # reference default seed, config.py:121


def quat_to_matrix_np(q: np.ndarray) -> np.ndarray:
    """Real-part-first quaternion (not necessarily unit) -> rotation matrix, same dtype as q."""
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    s2 = (2.0 / (q * q).sum(-1)).astype(q.dtype)
    m = np.stack([
        1 - s2 * (y * y + z * z), s2 * (x * y - z * w), s2 * (x * z + y * w),
        s2 * (x * y + z * w), 1 - s2 * (x * x + z * z), s2 * (y * z - x * w),
        s2 * (x * z - y * w), s2 * (y * z + x * w), 1 - s2 * (x * x + y * y),
    ], axis=-1)
    return m.reshape(q.shape[:-1] + (3, 3)).astype(q.dtype)


def uniform_rotations(n: int, seed: int = RD_SEED, dtype=np.float32) -> np.ndarray:
    """[n,3,3] Haar-uniform rotations (row-major), deterministic in (n, seed)."""
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((n, 4)).astype(dtype)
    nrm = np.sqrt((q * q).sum(-1))
    q = q / np.copysign(nrm, q[:, 0])[:, None]
    return np.ascontiguousarray(quat_to_matrix_np(q.astype(dtype)))


def features(n: int, dim: int, seed: int = RD_SEED + 1, dtype=np.float32) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n, dim)).astype(dtype)


REGIMES = {
    # name: (gain on Mobius conditioner fc_last, sigma of the 4x4 affine perturbation)
    "default": (1.0, 1e-3),
    "trained": (8.0, 0.2),
    # the "trained" weights moved to a random point of every conditioner MLP's ReLU-rescaling orbit (per hidden layer 2^U(-8,4), per unit
    # another 2^U(-2,2); flow/condition.py:24-30 computes the same function there up to fp32 rounding) and with the feature inputs of every
    # fc_first divided by FEATURE_SCALE while the features themselves are multiplied by it: what a checkpoint trained without weight
    # decay on un-normalised backbone features may look like.  Adversarial for arithmetic with an absolute floor.
    "imbalanced": (8.0, 0.2),
}
FEATURE_SCALE = {"imbalanced": 30.0}


def feature_scale(regime: str) -> float:
    """Factor the benchmark / fixture features of this weight regime are multiplied by (1 except for "imbalanced")."""
    return FEATURE_SCALE.get(regime, 1.0)


def _apply_relu_orbit(weights: dict, seed: int, fscale: float) -> None:
    """In place: every ConditionalTransform (keys <prefix>.fc_first / .layers.{1,3,5} / .fc_last) -> x0 * c0, x1 * c1, x2 * c2, x3 * c0
    (the residual x0 + x3 ties them), fc_last / c0; feature columns of fc_first / fscale."""
    prefixes = sorted(k[: -len(".fc_first.weight")] for k in weights if k.endswith(".fc_first.weight"))
    for idx, pre in enumerate(prefixes):
        rng = np.random.default_rng([seed, 7919, idx])
        c = (2.0 ** (rng.uniform(-8.0, 4.0, (3, 1)) + rng.uniform(-2.0, 2.0, (3, 64)))).astype(np.float32)
        c0, c1, c2 = c[0], c[1], c[2]
        w0 = weights[pre + ".fc_first.weight"]
        yo = 3 if pre.endswith(".conditioner") else 0
        w0[:, yo:] /= np.float32(fscale)
        w0 *= c0[:, None]
        weights[pre + ".fc_first.bias"] *= c0
        for j, (co, ci) in zip((1, 3, 5), ((c1, c0), (c2, c1), (c0, c2))):
            weights[pre + f".layers.{j}.weight"] *= co[:, None] / ci[None, :]
            weights[pre + f".layers.{j}.bias"] *= co
        weights[pre + ".fc_last.weight"] /= c0[None, :]


def fill_state_dict(shapes: dict, seed: int = 0, regime: str = "default") -> dict:
    """shapes: {state_dict key: shape tuple}.  Returns {key: float32 ndarray} filled by the recipe.

    Keys are visited in sorted order and each draws ``standard_normal(shape)`` from one shared stream, so two
    implementations whose state-dicts have the same keys/shapes get bit-identical weights.
    """
    gain, sigma = REGIMES[regime]
    rng = np.random.default_rng(seed)
    out = {}
    for key in sorted(shapes):
        shape = tuple(int(s) for s in shapes[key])
        z = rng.standard_normal(shape).astype(np.float32)
        if key.endswith(".mat") or key.endswith(".rot"):   # Uncondition16Trans.mat / UnconditionRot.rot [1,4,4]; 3x3 / 6x6 ablations
            val = np.eye(shape[-1], dtype=np.float32).reshape((1,) * (len(shape) - 2) + (shape[-1], shape[-1])) + np.float32(sigma) * z
        elif key.endswith(".net.w_p") or key.endswith(".net.u_mask") or key.endswith(".net.l_mask") or key.endswith(".net.l_eye") \
                or key.endswith(".net.s_sign"):             # ConditionLU buffers (flow/squeezetrans.py:110-114): same fixed structure
            d = shape[-1]
            name = key.rsplit(".", 1)[1]
            val = {"w_p": np.eye(d, dtype=np.float32)[[1, 0] + list(range(2, d))], "u_mask": np.triu(np.ones((d, d), dtype=np.float32), 1),
                   "l_mask": np.triu(np.ones((d, d), dtype=np.float32), 1).T, "l_eye": np.eye(d, dtype=np.float32),
                   "s_sign": np.array([1, -1, 1, 1], dtype=np.float32)[:d]}[name]
        elif key.endswith(".mat.w_p"):                   # UnconditionLU buffers: fixed, valid structure (the draw is discarded)
            d = shape[-1]
            val = np.eye(d, dtype=np.float32)[[1, 0] + list(range(2, d))]
        elif key.endswith(".mat.u_mask"):
            val = np.triu(np.ones(shape, dtype=np.float32), 1)
        elif key.endswith(".mat.l_mask"):
            val = np.triu(np.ones(shape, dtype=np.float32), 1).T
        elif key.endswith(".mat.l_eye"):
            val = np.eye(shape[-1], dtype=np.float32)
        elif key.endswith(".mat.s_sign"):
            val = np.array([1, -1, 1, 1], dtype=np.float32)[: shape[-1]]
        elif key.endswith(".mat.w_l") or key.endswith(".mat.w_u"):
            val = np.float32(max(sigma, 0.05)) * z
        elif key.endswith(".mat.w_s"):
            val = np.float32(0.2) * z
        elif key.endswith(".weight"):
            fan_in = shape[-1]
            val = z / np.float32(np.sqrt(fan_in))
            if "conditioner.fc_last" in key:
                val = val * np.float32(gain)
            elif "net.fc_last" in key:                 # Condition16Trans / ConditionLU nets: keep the per-sample matrix well conditioned
                val = val * np.float32(0.2)
        elif key.endswith(".bias"):
            val = np.float32(0.1) * z
            if "conditioner.fc_last" in key:
                val = val * np.float32(gain)
            elif "net.fc_last" in key:
                val = val * np.float32(0.2)
        else:
            raise KeyError(f"recipe has no rule for state-dict key {key!r}")
        out[key] = np.ascontiguousarray(val.astype(np.float32))
    if regime == "imbalanced":
        _apply_relu_orbit(out, seed, feature_scale(regime))
    return out


def fisher_A(kind: str = "diag531") -> np.ndarray:
    """Matrix-Fisher parameter used by the benchmark configs: A = diag(5,3,1), shape [1,3,3]."""
    if kind == "diag531":
        return np.diag(np.array([5.0, 3.0, 1.0], dtype=np.float32))[None]
    if kind == "tilted":
        rng = np.random.default_rng(7)
        u = uniform_rotations(2, seed=11)
        s = np.diag(np.array([6.0, 2.5, -0.75], dtype=np.float32))
        return (u[0] @ s @ u[1].T).astype(np.float32)[None] + 0 * rng.standard_normal(1).astype(np.float32)
    raise KeyError(kind)
