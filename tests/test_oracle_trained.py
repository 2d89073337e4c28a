"""CPU: the oracle against the fixtures that hold REAL trained weights -- checkpoints written by the reference's Flow under torch.optim.Adam
(tests/golden/make_trained.py; agent.py:23-28,75-92,132-151) -- and against the reference's own 20-step Adam trajectory.
Closes SURVEY 8(f) rank 1 on the oracle side: the restatement is pinned on saturated, peaked-density weights, not only on recipes."""
import numpy as np
import pytest
import torch

from oracle import flow_oracle as orc
from rotationnormflow_amd import synth
from tests.golden.trained_cases import TRAINED, TRAJ
from tests.trained_helpers import load_traj, load_trained


@pytest.mark.parametrize("name", list(TRAINED))
def test_checkpoint_layout_and_oracle_on_trained_weights(name):
    cfg, ckpt, w, fx, spec = load_trained(name)
    raw = torch.load(ckpt, map_location="cpu", weights_only=True)                      # plain tensors only: data, no code
    assert set(raw) == {"clock", "flow_state_dict"} and raw["clock"]["iteration"] == spec["steps"]      # agent.py:132-151
    shapes = orc.state_shapes(cfg)
    assert sorted(w) == sorted(shapes) and all(tuple(w[k].shape) == tuple(shapes[k]) for k in w)
    # these weights ARE trained: the flow puts the held-out test set at a mean log-likelihood far above the uniform density's 0
    assert float(fx["mean_ll64"]) > 4.0 and fx["curve"][-50:].mean() < -4.0 and fx["curve"][:5].mean() > -1.0
    feat = fx["test_feat"] if "test_feat" in fx else None
    R, ldj = orc.flow_forward(cfg, w, fx["test_rot"], feat, torch.float64)
    assert np.abs(ldj.numpy() - fx["ldj64"]).max() < 1e-10 and np.abs(R.numpy() - fx["rot64"]).max() < 1e-11
    assert abs(float(ldj.mean()) - float(fx["mean_ll64"])) < 1e-12                     # eval_uncondition.py:43-45
    m = fx["base_rot"].shape[0]
    Ri, li = orc.flow_inverse(cfg, w, fx["base_rot"], None if feat is None else feat[:m], torch.float64)
    assert np.abs(li.numpy() - fx["inv_ldj64"]).max() < 1e-9 and np.abs(Ri.numpy() - fx["inv_rot64"]).max() < 1e-10
    # fp32 oracle: inside the reference's own fp32 noise
    _, l32 = orc.flow_forward(cfg, w, fx["test_rot"], feat, torch.float32)
    noise = np.abs(fx["ldj32"].astype(np.float64) - fx["ldj64"])
    err = np.abs(l32.double().numpy() - fx["ldj64"])
    assert err.mean() <= 2 * noise.mean() + 1e-6 and err.max() <= 4 * noise.max() + 1e-5


@pytest.mark.parametrize("name", list(TRAJ))
def test_oracle_reproduces_the_reference_adam_trajectory(name):
    """20 steps of torch.optim.Adam on mean(-ldj) through the oracle's autograd graph (fp64) land on the reference's own fp64 trajectory:
    the same losses step by step and the same parameter update of every tensor."""
    cfg, fx, spec = load_traj(name)
    w0 = synth.fill_state_dict(orc.state_shapes(cfg), seed=spec["wseed"], regime=spec["regime"])
    p = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in w0.items()}
    opt = torch.optim.Adam(list(p.values()), spec["lr"])
    for it in range(spec["steps"]):
        batch = torch.from_numpy(fx["rot"][it * spec["batch"]: (it + 1) * spec["batch"]]).double()
        _, ldj = orc.flow_forward(cfg, p, batch, None, dtype=torch.float64, grad=True)
        loss = (-ldj).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        assert abs(float(loss.detach()) - fx["loss64"][it]) < 1e-9, it
    for k in p:
        dw = p[k].detach().numpy() - w0[k].astype(np.float64)
        want = fx["dw64:" + k].astype(np.float64)
        assert np.abs(dw - want).max() <= 2e-7 * max(1.0, np.abs(want).max()) + 1e-9, k     # (the fixture stores the updates as float32)
