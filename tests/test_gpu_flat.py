"""GPU (-m gpu): the flattened-parameter mode of a Flow (Flow.flatten_parameters; what get_flow hands the reference's unedited drivers,
agent.py:20-23) -- one nn.Parameter behind the reference's state-dict keys.  Same kernels, same results; one autograd leaf, one optimizer tensor."""
import contextlib
import io

import numpy as np
import pytest
import torch

from oracle import flow_oracle as orc
from rotationnormflow_amd import harness, make_config, synth
from rotationnormflow_amd.flow.flow import Flow, get_flow
from tests.trained_helpers import load_traj

pytestmark = pytest.mark.gpu

STRUCTS = {
    "uncond": dict(layers=3, segments=16),
    "cond": dict(layers=2, segments=16, condition=1, feature_dim=24, rot="16UnTrans", frequent_permute=1, last_affine=1, first_affine=0),
}


def _pair(cfg, seed=5, regime="trained"):
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=seed, regime=regime)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    with contextlib.redirect_stdout(io.StringIO()):
        classic, flat = Flow(cfg), get_flow(cfg)
    classic.load_state_dict(sd)
    flat.load_state_dict(sd)
    assert flat.is_flat
    return w, classic.cuda(), flat.cuda()


@pytest.mark.parametrize("name", list(STRUCTS))
def test_flat_flow_gives_the_same_results_and_gradients(name):
    cfg = make_config(None, **STRUCTS[name])
    w, classic, flat = _pair(cfg)
    n = 300
    R = torch.from_numpy(synth.uniform_rotations(n, seed=1)).cuda()
    feat = torch.from_numpy(synth.features(n, 24, seed=2)).cuda() if cfg.condition else None
    # after .cuda() the per-layer views still alias the one parameter
    assert flat.layers[0 if not cfg.condition else 1].conditioner.fc_first.weight.data_ptr() >= flat._flat.data_ptr()
    for mode in ("eval", "train"):
        getattr(classic, mode)()
        getattr(flat, mode)()
        with torch.no_grad():
            a, b = classic.log_prob(R, feat), flat.log_prob(R, feat)
            ai, bi = classic.inverse(R, feat), flat.inverse(R, feat)
        assert torch.equal(a["logp"], b["logp"]) and torch.equal(ai[0], bi[0]) and torch.equal(ai[1], bi[1]), mode
    g = torch.from_numpy(np.random.default_rng(3).standard_normal(n).astype(np.float32)).cuda()
    grads = []
    for fl in (classic, flat):
        fl.train()
        fl.zero_grad()
        Rq = R.clone().requires_grad_(True)
        rot, ldj = fl(Rq, feat)
        ((ldj * g).sum() + rot.square().sum()).backward()
        grads.append((fl.named_parameter_gradients(), Rq.grad))
    (gc, rc), (gf, rf) = grads
    assert sorted(gc) == sorted(gf) == sorted(w)
    assert flat._flat.grad is not None and flat._flat.grad.numel() == flat._flat.numel()
    assert all(p.grad is None for p in flat.layers.parameters())          # (there are none: the per-layer tensors are buffers)
    for k in gc:                                                           # float atomics order: equal to rounding, not bit for bit
        assert torch.allclose(gc[k], gf[k], rtol=2e-4, atol=1e-6 * float(gc[k].abs().max() + 1e-12) + 1e-9), k
    assert torch.allclose(rc, rf, rtol=1e-4, atol=1e-6)


def test_reference_driver_order_plain_adam_follows_the_reference_trajectory():
    """agent.py:20-28 + 49-52: get_flow -> DataParallel -> Adam(flow.parameters(), lr) on the CPU module, THEN .cuda() at the first forward;
    torch's default Adam.  The flattened flow must follow the reference's own 20-step fp64 trajectory (tests/golden/traj_c1.npz) like the
    per-tensor flow does (tests/test_gpu_trained.py), with the optimizer stepping ONE tensor."""
    cfg, fx, spec = load_traj("traj_c1")
    w0 = synth.fill_state_dict(orc.state_shapes(cfg), seed=spec["wseed"], regime=spec["regime"])
    with contextlib.redirect_stdout(io.StringIO()):
        fl = get_flow(cfg)
    fl.load_state_dict({k: torch.from_numpy(v) for k, v in w0.items()})
    dp = torch.nn.DataParallel(fl)
    opt = torch.optim.Adam(dp.parameters(), spec["lr"])                     # built while the module is still on the CPU
    assert len(opt.param_groups[0]["params"]) == 1
    B = spec["batch"]
    R = torch.from_numpy(fx["rot"])
    losses = []
    for it in range(spec["steps"]):
        flow = dp.cuda()                                                    # agent.py:52 (every iteration)
        flow.train()
        _, ldj = flow(R[it * B:(it + 1) * B].cuda(), None)
        loss = (-ldj).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    losses = np.array(losses)
    ref_noise = np.abs(fx["loss32"] - fx["loss64"])
    assert np.abs(losses - fx["loss64"]).max() <= 4 * ref_noise.max() + 2e-6
    sd = {k: v.detach().cpu().double().numpy() for k, v in fl.state_dict().items()}
    for k, v in sd.items():
        want = fx["dw64:" + k].astype(np.float64)
        err = np.linalg.norm((v - w0[k].astype(np.float64)) - want)
        assert err <= 4.0 * float(fx["ref32_err:" + k]) + 2e-3 * np.linalg.norm(want) + 1e-7, k
    # evaluation right after training sees the stepped parameters (pack cache keyed on the flat parameter)
    fl.eval()
    Rt = R[:512].cuda()
    with torch.no_grad():
        got = fl.log_prob(Rt)["logp"].cpu().double()
    want, _ = orc.log_prob(cfg, {k: v for k, v in sd.items()}, R[:512].numpy(), None, None, torch.float64)
    assert (got - want).abs().max() < 5e-5
    # checkpoint in the reference's layout, optimizer state included
    ck = harness.expand_optimizer_state(fl, opt.state_dict())
    assert len(ck["state"]) == len(w0) and float(ck["state"][0]["step"]) == spec["steps"]


def test_flat_flow_in_a_hip_graph_and_layer_called_on_its_own():
    cfg = make_config(None, layers=3, segments=16)
    w, classic, flat = _pair(cfg, seed=41, regime="default")
    R = torch.from_numpy(synth.uniform_rotations(256, seed=9)).cuda()
    flat.train()
    classic.train()
    opt_f = torch.optim.Adam(flat.parameters(), 1e-3, fused=True, capturable=True)
    opt_c = torch.optim.Adam(classic.parameters(), 1e-3, fused=True)
    step = harness.GraphedTrainStep(flat, opt_f, (256, 3, 3))
    for it in range(6):
        lf = float(step(R).detach())
        _, ldj = classic(R)
        lc = (-ldj).mean()
        opt_c.zero_grad()
        lc.backward()
        opt_c.step()
        assert abs(lf - float(lc.detach())) < 2e-5 * max(1.0, abs(lf)), (it, lf, float(lc))
    # a layer of the flattened flow called on its own (reference API: layer(rotation, permute_row, feature)) evaluates the CURRENT weights
    flat.eval()
    classic.eval()
    perm = torch.tensor([0, 1, 2])
    with torch.no_grad():
        a = flat.layers[0](R, perm)
        flat._flat.mul_(1.01)                                              # an "optimizer step"
        b = flat.layers[0](R, perm)
    assert not torch.equal(a[1], b[1])
