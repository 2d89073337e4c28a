"""Helpers of the trained-weights tests: fixtures of tests/golden/make_trained.py (checkpoints the reference's own Adam loop produced)."""
import os

import numpy as np

from rotationnormflow_amd import harness
from rotationnormflow_amd.configs import make_config
from tests.golden.trained_cases import TRAINED, TRAJ

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_trained(name):
    """-> (cfg, checkpoint path, weights {key: f32 ndarray}, fixture npz, spec)"""
    spec = TRAINED[name]
    cfg = make_config(**spec["cfg"])
    ckpt = os.path.join(GOLDEN, name + ".pth")
    sd = harness.load_reference_checkpoint(ckpt)
    return cfg, ckpt, {k: v.numpy() for k, v in sd.items()}, np.load(os.path.join(GOLDEN, name + ".npz")), spec


def load_traj(name):
    spec = TRAJ[name]
    return make_config(**spec["cfg"]), np.load(os.path.join(GOLDEN, name + ".npz")), spec
