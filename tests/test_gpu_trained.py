"""GPU (-m gpu): the HIP path on weights the REFERENCE'S OWN TRAINING produced (tests/golden/make_trained.py: the reference's Flow under
torch.optim.Adam on sharp matrix-Fisher mixtures; agent.py:23-28,75-92), loaded from checkpoints in Agent.save_ckpt's layout through the
harness (agent.py:132-151,171-198), and on the reference's own 20-step Adam trajectory.  SURVEY 8(f) rank 1: real-weights parity."""
import numpy as np
import pytest
import torch

from rotationnormflow_amd import harness, runtime, synth
from tests.golden.trained_cases import TRAINED, TRAJ
from tests.gpu_helpers import product_flow
from tests.trained_helpers import load_traj, load_trained

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["f16x2", "fp32"])
def precision(request):
    old = runtime.get_precision()
    runtime.set_precision(request.param)
    yield request.param
    runtime.set_precision(old)


@pytest.mark.parametrize("name", list(TRAINED))
def test_trained_checkpoint_eval_statistic_forward_and_inverse(name, precision):
    cfg, ckpt, w, fx, spec = load_trained(name)
    flow = harness.build_flow_from_checkpoint(cfg, ckpt)
    R = torch.from_numpy(fx["test_rot"]).cuda()
    feat = torch.from_numpy(fx["test_feat"]).cuda() if "test_feat" in fx else None
    with torch.no_grad():
        Rt, ldj = flow(R, feat)
    packed = flow._packed(R.device, feat)
    assert packed.precision == precision                                           # trained weights pass the pack-time audit: no silent fp32 fallback
    print(f"{name} [{precision}]: pack audit {packed.audit:.2e}")
    if precision == "f16x2":
        assert 0.0 < packed.audit < 4e-6                                           # refused above 4e-6
        # ... and the launch guard stays quiet on them (round 4: segment weights up to s = 250 no longer overflow the lean softplus)
        assert not runtime.fallback_fired(R.device)
    ldj = ldj.cpu().double().numpy()
    Rt = Rt.cpu().double().numpy()
    # the reference's eval statistic (eval_uncondition.py:43-45: mean over the test set of ldj + base log-prob; uniform base)
    if feat is None:
        got = harness.mean_log_likelihood(flow, torch.from_numpy(fx["test_rot"]), batch_size=700)          # ragged batches
        assert abs(got - float(fx["mean_ll64"])) < 1e-5
    assert abs(ldj.mean() - float(fx["mean_ll64"])) < 1e-5
    noise = np.abs(fx["ldj32"].astype(np.float64) - fx["ldj64"])
    err = np.abs(ldj - fx["ldj64"])
    assert err.mean() <= 2 * noise.mean() + 2e-6 and err.max() <= 4 * noise.max() + 2e-5          # the gates of tests/test_gpu_parity.py
    # ... and their per-sample form (see test_forward_matches_reference_golden for what an independent fp32 evaluation can meet)
    per = np.maximum(1e-5, 2 * noise)
    frac, excess = float(np.mean(err <= per)), float(np.max(err - per))
    p99_32, p99_ref = float(np.quantile(np.abs(ldj - fx["ldj32"].astype(np.float64)), 0.99)), float(np.quantile(noise, 0.99))
    print(f"{name} [{precision}]: per-sample pass {frac:.4f}, worst excess {excess:.1e}, p99 vs ref32 {p99_32:.1e} (reference's own {p99_ref:.1e})")
    if precision == "f16x2":
        assert frac >= 0.985 and excess <= noise.max() + 1e-5 and p99_32 <= p99_ref + 1e-5, (frac, excess, p99_32, p99_ref)
    else:
        assert frac >= 0.88 and excess <= 2 * noise.max() + 1e-5 and p99_32 <= 1.5 * p99_ref + 1e-5, (frac, excess, p99_32, p99_ref)
    rnoise = np.abs(fx["rot32"].astype(np.float64) - fx["rot64"]).max()
    assert np.abs(Rt - fx["rot64"]).max() <= 4 * rnoise + 1e-5
    # inverse pass on base samples (agent.py:238-263), in bisection cells
    m = fx["base_rot"].shape[0]
    with torch.no_grad():
        Ri, li = flow.inverse(torch.from_numpy(fx["base_rot"]).cuda(), None if feat is None else feat[:m])
    Ri, li = Ri.cpu().double().numpy(), li.cpu().double().numpy()
    inoise = np.abs(fx["inv_ldj32"].astype(np.float64) - fx["inv_ldj64"])
    irn = np.abs(fx["inv_rot32"].astype(np.float64) - fx["inv_rot64"]).reshape(m, -1).max(1)
    ierr = np.abs(li - fx["inv_ldj64"])
    irerr = np.abs(Ri - fx["inv_rot64"]).reshape(m, -1).max(1)
    cell = np.pi / 2 ** 14
    assert ierr.mean() <= 3 * inoise.mean() + 1e-5 and irerr.mean() <= 3 * irn.mean() + 1e-5
    # trained weights stretch a cell more than the recipes do: |d ldj / d theta| reaches a few tens at the sharp modes
    assert irerr.max() <= 2.0 * cell + irn.max() and ierr.max() <= 6 * cell * max(1.0, np.abs(fx["inv_ldj64"]).max()) + inoise.max()
    assert np.mean(irerr > 0.5 * cell) <= max(0.01, np.mean(irn > 0.5 * cell)) + 0.005


@pytest.mark.parametrize("name", list(TRAJ))
@pytest.mark.parametrize("graph", [False, True])
def test_hip_training_follows_the_reference_adam_trajectory(name, graph):
    """20 iterations of the reference's training step (agent.py:75-92: loss = mean(-ldjs), zero_grad, backward, Adam(lr).step()) on the HIP
    training path -- eager with the reference's plain torch.optim.Adam, and replayed as a HIP graph -- against the reference's own fp64
    trajectory, step by step; gate: as close to it as the reference's own fp32 run (stored per tensor in the fixture)."""
    cfg, fx, spec = load_traj(name)
    from oracle import flow_oracle as orc
    w0 = synth.fill_state_dict(orc.state_shapes(cfg), seed=spec["wseed"], regime=spec["regime"])
    fl = product_flow(cfg, w0).train()
    B = spec["batch"]
    R = torch.from_numpy(fx["rot"]).cuda()
    losses = []
    if graph:
        # the optimizer harness.train_uncondition builds for graphed training.  (PyTorch's FOREACH capturable Adam -- capturable=True without
        # fused -- drifts from the reference's trajectory by 1e-6 of loss per step, eagerly too: its device-side fp32 bias corrections, not
        # this library; measured in round 4, tools/_build/traj_diag.py.  The fused kernel follows the reference at its own fp32 noise.)
        opt = torch.optim.Adam(fl.parameters(), spec["lr"], capturable=True, fused=True)
        step = harness.GraphedTrainStep(fl, opt, (B, 3, 3))
        for it in range(spec["steps"]):
            losses.append(float(step(R[it * B:(it + 1) * B]).detach()))
    else:
        opt = torch.optim.Adam(fl.parameters(), spec["lr"])                     # agent.py:23
        for it in range(spec["steps"]):
            _, ldj = fl(R[it * B:(it + 1) * B])
            loss = (-ldj).mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    losses = np.array(losses)
    ref_noise = np.abs(fx["loss32"] - fx["loss64"])
    assert np.abs(losses - fx["loss64"]).max() <= 4 * ref_noise.max() + 2e-6, (losses - fx["loss64"])
    assert losses[-1] < losses[0] - 1.0                                         # it trains: 0.06 -> -2.05
    # parameter updates after 20 steps: per tensor, no farther from the fp64 truth than a few times the reference's own fp32 run
    sd = {k: v.detach().cpu().double().numpy() for k, v in fl.state_dict().items()}
    worst = 0.0
    for k, v in sd.items():
        want = fx["dw64:" + k].astype(np.float64)
        err = np.linalg.norm((v - w0[k].astype(np.float64)) - want)
        tol = 4.0 * float(fx["ref32_err:" + k]) + 2e-3 * np.linalg.norm(want) + 1e-7
        worst = max(worst, err / tol)
        assert err <= tol, (k, err, float(fx["ref32_err:" + k]), np.linalg.norm(want))
    print(f"{name} graph={graph}: loss err max {np.abs(losses - fx['loss64']).max():.2e} (reference fp32: {ref_noise.max():.2e}); worst update err / tol {worst:.2f}")


@pytest.mark.parametrize("name", list(TRAINED))
def test_trained_checkpoint_at_full_batch_runs_on_the_fast_path(name):
    """The checkpoints the reference's own training produced at BASELINE batch size (2^20 rotations): the default split-precision kernels
    carry them WITHOUT the exact-fp32 re-run (round 3's lean softplus overflowed on these weights and every launch was re-run), shards of
    any size reproduce their rows bit for bit, inverse(forward(R)) returns R to the bisection cells of the stack, and the mean NLL over 2^20
    uniform rotations is consistent with a normalised density (E_uniform[exp(log p)] = 1)."""
    from rotationnormflow_amd.dist import shard_bounds
    cfg, ckpt, w, fx, spec = load_trained(name)
    flow = harness.build_flow_from_checkpoint(cfg, ckpt)
    n = 1 << 20
    R = torch.from_numpy(synth.uniform_rotations(n, seed=77)).cuda()
    feat = None
    if cfg.condition:                                              # features of the three classes the flow was trained on (+ noise), one row per rotation
        base = torch.from_numpy(fx["test_feat"]).cuda()
        feat = base[torch.arange(n, device="cuda") % base.shape[0]].contiguous()
    with torch.no_grad():
        full = flow.log_prob(R, feat)
    assert runtime.get_precision() == "f16x2" and flow._packed(R.device, feat).precision == "f16x2"
    assert not runtime.fallback_fired(R.device)
    lp = full["logp"]
    assert torch.isfinite(lp).all()
    for r in (0, 3, 7):
        lo, hi = shard_bounds(n, r, 8)
        with torch.no_grad():
            part = flow.log_prob(R[lo:hi], None if feat is None else feat[lo:hi])["logp"]
        assert torch.equal(part, lp[lo:hi])
    with torch.no_grad():
        small = flow.log_prob(R[:3000].contiguous(), None if feat is None else feat[:3000].contiguous())["logp"]
    assert torch.equal(small, lp[:3000])
    # a normalised density: the Monte-Carlo estimate of its integral over SO(3) (Haar measure normalised to 1) is 1.  The trained densities
    # are sharp (modes of kappa = 20), so the estimator's own spread is large: checked within its standard error
    p = torch.exp(lp.double())
    mass, se = p.mean().item(), (p.std() / np.sqrt(n)).item()
    assert abs(mass - 1.0) < 6 * se + 0.02, (mass, se)
    # round trip on a slice
    m = 1 << 16
    with torch.no_grad():
        Rt, ldj = flow(R[:m], None if feat is None else feat[:m])
        Rb, ldjb = flow.inverse(Rt, None if feat is None else feat[:m])
    # The inverse lands on the reference's bisection grid (cells of pi / 2^15 per layer) and the layers behind a cell stretch it by their
    # Jacobians -- uniform rotations are far from the modes these densities were trained on, where the stretch is largest -- so the gate is
    # the REFERENCE's own round trip (the oracle in fp32 arithmetic, forward then inverse) on the first rows, not a constant: 8-layer flows
    # sit at 2e-4, the 24-layer SYMSOL structure at 1.7e-3.
    from oracle import flow_oracle as orc
    k = 256
    Rk = R[:k].cpu().numpy()
    fk = None if feat is None else feat[:k].cpu().numpy()
    oR, ol = orc.flow_forward(cfg, w, Rk, fk, dtype=torch.float32)
    oRb, olb = orc.flow_inverse(cfg, w, oR.numpy(), fk, dtype=torch.float32)
    ref_rt = (oRb - torch.from_numpy(Rk)).abs().mean().item()
    ref_ld = (ol + olb).abs().mean().item()
    got_rt, got_ld = (Rb[:k] - R[:k]).abs().mean().item(), (ldj[:k] + ldjb[:k]).abs().mean().item()
    print(f"{name}: round trip {got_rt:.2e} (reference arithmetic {ref_rt:.2e}), log-det {got_ld:.2e} ({ref_ld:.2e}); all 2^16 rows {(Rb - R[:m]).abs().mean().item():.2e}")
    assert got_rt <= 2.0 * ref_rt + 1e-4 and got_ld <= 2.0 * ref_ld + 5e-4, (got_rt, ref_rt, got_ld, ref_ld)
    assert (Rb - R[:m]).abs().mean().item() <= 3.0 * ref_rt + 2e-4 and (ldj + ldjb).abs().mean().item() <= 3.0 * ref_ld + 1e-3


def test_rootfinder_first_pass_order_changes_speed_not_cells():
    """Round 6: the fourth-order first pass of the inverse root finder (Flow.set_rootfinder_order; the SYMSOL checkpoint's sidecar asks for
    it) returns the rotations of the third-order one -- both land on the reference's bisection grid, so rows differ only where a root sits
    within rounding of a cell boundary (then by one cell of one layer) -- and is a property of the flow: rows do not depend on the batch."""
    cfg, ckpt, w, fx, spec = load_trained("trained_c4")
    flow = harness.build_flow_from_checkpoint(cfg, ckpt)
    assert flow._rnf_rf_order == 4                                  # tests/golden/trained_c4.pth.rnf.json
    n = 1 << 15
    R = torch.from_numpy(synth.uniform_rotations(n, seed=5)).cuda()
    base = torch.from_numpy(fx["test_feat"]).cuda()
    feat = base[torch.arange(n, device="cuda") % base.shape[0]].contiguous()
    out = {}
    with torch.no_grad():
        for order in (4, 3, None):
            flow.set_rootfinder_order(order)
            out[order] = flow.inverse(R, feat)
            part = flow.inverse(R[1000:3000].contiguous(), feat[1000:3000].contiguous())
            assert torch.equal(part[0], out[order][0][1000:3000]) and torch.equal(part[1], out[order][1][1000:3000])
    assert torch.equal(out[3][0], out[None][0]) and torch.equal(out[3][1], out[None][1])          # the default IS the third order
    same = (out[4][0] == out[3][0]).flatten(1).all(1)
    assert same.double().mean().item() > 0.97, same.double().mean().item()
    # 24 Moebius layers, a cell of pi / 2^14 each, stretched by the layers behind it: the rows that differ do so at the scale of the
    # build's own round-trip error (test above: 1.7e-3 mean for this structure)
    assert (out[4][0] - out[3][0]).abs().max().item() < 0.05 and (out[4][1] - out[3][1]).abs().mean().item() < 2e-4


@pytest.mark.parametrize("flat", [True, False])
def test_agent_checkpoint_path_resumes_the_reference_trajectory(tmp_path, flat, monkeypatch):
    """The reference's checkpoint path on the GPU, statement by statement (tests/agent_replay.py: Agent.__init__ agent.py:20-28, load_ckpt
    agent.py:171-198 incl. the unconditional Adam ``load_state_dict``, train_func agent.py:75-92, save_ckpt agent.py:132-152 incl.
    ``.module.cpu().state_dict()`` -> ``.cuda()``): the checkpoint the REFERENCE wrote after step 10 of traj_c1 (its own Flow + Adam, per-tensor
    optimizer state) is loaded by the default drop-in (``get_flow``: flattened parameters), trained for 5 steps, saved, loaded by a fresh agent,
    trained for 5 more -- and lands on the reference's own step-20 weights and losses within the trajectory gate of the test above."""
    import contextlib
    import io
    import os
    from oracle import flow_oracle as orc
    from rotationnormflow_amd.flow.flow import get_flow
    from tests.agent_replay import ReplayedAgent
    from tests.trained_helpers import GOLDEN
    monkeypatch.setenv("RNF_FLAT_PARAMS", "1" if flat else "0")
    cfg, fx, spec = load_traj("traj_c1")

    def quiet_get_flow(c):
        with contextlib.redirect_stdout(io.StringIO()):
            return get_flow(c)
    w0 = synth.fill_state_dict(orc.state_shapes(cfg), seed=spec["wseed"], regime=spec["regime"])
    B = spec["batch"]
    R = torch.from_numpy(fx["rot"])
    agent = ReplayedAgent(cfg, quiet_get_flow, "cuda", lr=spec["lr"])
    assert agent.flow.module.is_flat == flat
    agent.load_ckpt(os.path.join(GOLDEN, "traj_c1_step10.pth"))
    assert agent.clock["iteration"] == 10
    losses = [agent.train_func(R[it * B:(it + 1) * B]) for it in range(10, 15)]
    agent.save_ckpt(tmp_path / "mid.pth")                                                  # .cpu() ... .cuda() around the state_dict
    mid = torch.load(tmp_path / "mid.pth", map_location="cpu", weights_only=False)
    assert len(mid["optimizer_flow_state_dict"]["state"]) == len(w0) and float(mid["optimizer_flow_state_dict"]["state"][0]["step"]) == 15.0
    assert next(agent.flow.parameters()).is_cuda
    agent = ReplayedAgent(cfg, quiet_get_flow, "cuda", lr=spec["lr"])                      # a new process would start here
    agent.load_ckpt(tmp_path / "mid.pth")
    assert agent.clock["iteration"] == 15
    losses += [agent.train_func(R[it * B:(it + 1) * B]) for it in range(15, 20)]
    losses = np.array(losses)
    ref_noise = np.abs(fx["loss32"] - fx["loss64"])
    assert np.abs(losses - fx["loss64"][10:]).max() <= 4 * ref_noise.max() + 2e-6, (losses - fx["loss64"][10:])
    sd = {k: v.detach().cpu().double().numpy() for k, v in agent.flow.module.state_dict().items()}
    worst = 0.0
    for k, v in sd.items():
        want = fx["dw64:" + k].astype(np.float64)
        err = np.linalg.norm((v - w0[k].astype(np.float64)) - want)
        tol = 4.0 * float(fx["ref32_err:" + k]) + 2e-3 * np.linalg.norm(want) + 1e-7
        worst = max(worst, err / tol)
        assert err <= tol, (k, err, float(fx["ref32_err:" + k]), np.linalg.norm(want))
    print(f"resume flat={flat}: loss err max {np.abs(losses - fx['loss64'][10:]).max():.2e}; worst update err / tol {worst:.2f}")
