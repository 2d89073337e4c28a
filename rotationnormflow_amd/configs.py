"""Flow configuration objects.

The reference builds its ``Flow`` from an attribute bag produced by configargparse (config.py:94-238); only the
attributes listed in SURVEY.md 8(b) are read on the hot path.  ``make_config`` returns such a bag with the
reference's defaults (settings/base.yml + config.py:117-166); ``PRESETS`` names the BASELINE.json configurations.
"""
from __future__ import annotations

from types import SimpleNamespace

_DEFAULTS = dict(
    dist="mobiusflow", layers=24, segments=64, rot="16Trans", lu=0,
    condition=0, feature_dim=None, embedding=0, embedding_dim=0,
    last_affine=0, first_affine=1, frequent_permute=0,
)

# BASELINE.json `configs` -> flow structure ("Layer stacks these configs produce")
PRESETS = {
    # settings/raw.yml
    "C1": dict(layers=8),
    "C2": dict(layers=24),
    "C3": dict(layers=24),
    # settings/symsol.yml --layers 24 --feature_dim 256
    "C4": dict(layers=24, condition=1, feature_dim=256, rot="16UnTrans",
               frequent_permute=1, last_affine=1, first_affine=0),
    # README.md:153  settings/symsol.yml --layers 42 --last_affine 0 --rot None
    "C5": dict(layers=42, condition=1, feature_dim=512, rot="None",
               frequent_permute=1, last_affine=0, first_affine=0),
    # unconditional variant of C5 (same Moebius-only stack without the feature input)
    "C5u": dict(layers=42, condition=0, rot="None", last_affine=0, first_affine=0),
}


def make_config(preset: str | None = None, **overrides) -> SimpleNamespace:
    kw = dict(_DEFAULTS)
    if preset is not None:
        kw.update(PRESETS[preset])
    kw.update(overrides)
    return SimpleNamespace(**kw)


def load_yaml_config(*paths, **overrides) -> SimpleNamespace:
    """Read the flow-relevant keys of the reference's settings/*.yml files (later files override earlier ones,
    as configargparse does with base.yml + --config, config.py:95-101)."""
    import yaml

    kw = dict(_DEFAULTS)
    for path in paths:
        with open(path) as fh:
            doc = yaml.safe_load(fh) or {}
        for key in _DEFAULTS:
            if key in doc:
                kw[key] = doc[key]
    kw.update(overrides)
    return SimpleNamespace(**kw)
