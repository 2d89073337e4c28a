"""Case table of tests/golden/make_trained.py (generator, reference side) and tests/test_gpu_trained.py / test_oracle_trained.py (checkers).

TRAINED: name -> flow config + the synthetic ``raw``-style target + the optimisation recipe.  TRAJ: short Adam trajectories from recipe weights.
"""
SYMSOL = dict(condition=1, rot="16UnTrans", frequent_permute=1, last_affine=1, first_affine=0)

TRAINED = {
    # BASELINE configs[0] structure (settings/raw.yml with --layers 8): two sharp matrix-Fisher modes (kappa 20: ~9 degrees per axis)
    "trained_c1": dict(cfg=dict(layers=8), n_modes=2, kappa=20.0, mode_seed=301, data_seed=302, init_seed=303,
                       n_train=16384, n_test=2048, n_inverse=512, steps=3000, batch=256, lr=1e-3),
    # 4-layer conditional flow, SYMSOL structure (Condition16Trans first, Uncondition16Trans between), F = 32: three feature classes, each a
    # two-mode target
    "trained_cond4": dict(cfg=dict(layers=4, feature_dim=32, **SYMSOL), n_classes=3, kappa=20.0, mode_seed=311, data_seed=312, init_seed=313,
                          n_train=16384, n_test=1024, n_inverse=512, steps=2500, batch=256, lr=1e-3),
}

TRAJ = {
    "traj_c1": dict(cfg=dict(layers=3, segments=32), n_modes=2, kappa=20.0, mode_seed=301, data_seed=322, wseed=323, regime="default",
                    steps=20, batch=256, lr=1e-3),
}
