"""rotationnormflow_amd -- MI355X-native SO(3) normalizing-flow density path (drop-in for PKU-EPIC/RotationNormFlow's
``flow.flow.Flow`` forward / inverse / log-prob, matrix-Fisher base log-density and mean-NLL reduction).

The per-sample math runs in hand-written HIP kernels for gfx950 behind the C ABI in include/rnf_hip.h; this package is
the Python host side that mirrors the reference's module API (same class names, constructor arguments, call
signatures and state-dict keys).  There is no CPU fallback.
"""
from .configs import make_config, load_yaml_config, PRESETS  # noqa: F401

__all__ = ["make_config", "load_yaml_config", "PRESETS", "install_as_reference_modules", "set_precision", "get_precision",
           "set_feature_calibration", "get_feature_calibration"]


def set_precision(name):
    """Arithmetic of the conditioner GEMMs: "f16x2" (default, split-precision fp16 MFMA), "bf16x3" (strict: 24-bit operands as three bf16
    terms on the bf16 MFMA -- no equalisation / calibration / range guard) or "fp32" (exact fp32-input MFMA)."""
    from . import runtime
    runtime.set_precision(name)


def get_precision():
    from . import runtime
    return runtime.get_precision()


def set_feature_calibration(mode):
    """Where the pack-time equalisation of conditional layers takes the feature scale from: "weights" (default: 1 unless fixed with
    ``Flow.set_feature_scale`` / ``calibrate_feature_scale`` / a checkpoint sidecar -- deterministic) or "first-batch" (measured on the first
    feature batch a parameter version is packed for, re-measured when the launch guard keeps firing).  Env: RNF_FEATURE_CALIBRATION."""
    from . import runtime
    runtime.set_feature_calibration(mode)


def get_feature_calibration():
    from . import runtime
    return runtime.get_feature_calibration()


def install_as_reference_modules():
    """Register this package's modules under the reference's import names (``flow.flow``, ``flow.mobiusflow``,
    ``flow.affineflow``, ``flow.squeezetrans``, ``flow.condition``, ``utils.fisher``) so that the reference's own
    scripts (agent.py:9-10: ``from flow.flow import Flow, get_flow``; ``from utils.fisher import MatrixFisherN``) pick
    up the HIP implementation unchanged.  Call it before importing the reference's ``agent``; see INTEGRATION.md."""
    import importlib
    import sys
    import types

    from . import flow as _flow_pkg

    sys.modules["flow"] = _flow_pkg
    for name in ("flow", "mobiusflow", "affineflow", "squeezetrans", "rottrans", "condition"):
        sys.modules[f"flow.{name}"] = importlib.import_module(f"{__name__}.flow.{name}")
    from .utils import fisher as _fisher

    # the reference's `utils` package holds other (out-of-scope) modules; only its `fisher` submodule is replaced
    sys.modules["utils.fisher"] = _fisher
    utils_pkg = sys.modules.get("utils")
    if utils_pkg is None:
        try:
            utils_pkg = importlib.import_module("utils")
        except ImportError:
            utils_pkg = types.ModuleType("utils")
            utils_pkg.__path__ = []
            sys.modules["utils"] = utils_pkg
    utils_pkg.fisher = _fisher
