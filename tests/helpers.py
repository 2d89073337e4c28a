"""Shared helpers for the parity tests: rebuild a golden case's inputs/weights from the recipe."""
import os
import zlib

import numpy as np

from oracle import flow_oracle as orc
from rotationnormflow_amd import synth
from rotationnormflow_amd.configs import make_config
from tests.golden.cases import CASES

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def load_case(name):
    """-> (cfg, weights{key: f32 ndarray}, R_in [n,3,3] f32, feature [n,F] f32 | None, fixture npz, spec)"""
    spec = CASES[name]
    cfg = make_config(**spec["cfg"])
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    shapes = orc.state_shapes(cfg)
    weights = synth.fill_state_dict(shapes, seed=spec["wseed"], regime=spec["regime"])
    R = synth.uniform_rotations(spec["n"], seed=spec["rseed"])
    feat = None
    if cfg.condition:
        feat = synth.features(spec["n"], int(fx["feature_dim"]), seed=spec["rseed"] + 1000) * np.float32(synth.feature_scale(spec["regime"]))
    return cfg, weights, R, feat, fx, spec
