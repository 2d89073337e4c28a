"""GPU: MatrixFisherN(A, norm_type)._log_prob differentiated w.r.t. the parameter matrix A (agent.py:57-65 keeps a predicted A in the
autograd graph) and the rotations, for both closed-form normaliser approximations: rnf_fisher_log_const_nt, rnf_fisher_log_prob,
rnf_fisher_log_prob_backward, rnf_fisher_log_prob_backward_param through the C ABI.

Checked against (1) the fixture of the reference's OWN autograd through torch.svd (tests/golden/fisher_grad.npz) and (2) autograd of the
fp64 oracle at sizes that exercise both accumulation paths of the kernel (one row for the whole batch; one row per sample; rows that
straddle the 256-sample steps).  Tolerance: 2e-5 of the largest entry of the gradient + the reference's own fp32-vs-fp64 difference."""
import os

import numpy as np
import pytest
import torch

from oracle import flow_oracle as orc
from rotationnormflow_amd import synth
from rotationnormflow_amd.utils.fisher import MatrixFisherN

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("norm_type", [0, 1])
def test_fisher_gradients_match_the_reference_fixture(golden_dir, norm_type):
    d = np.load(os.path.join(golden_dir, "fisher_grad.npz"))
    A = torch.from_numpy(d["A"]).float().cuda().requires_grad_(True)
    R = torch.from_numpy(d["R"]).float().cuda().requires_grad_(True)
    lp = MatrixFisherN(A, norm_type=norm_type)._log_prob(R)
    (lp * torch.from_numpy(d["g"]).float().cuda()).sum().backward()
    for got, key in ((lp.detach(), "logp"), (A.grad, "gA"), (R.grad.reshape(-1, 3, 3), "gR")):
        want, ref32 = d[f"{key}_t{norm_type}_64"], d[f"{key}_t{norm_type}_32"]
        tol = 2e-5 * np.abs(want).max() + 2 * np.abs(ref32 - want).max()
        assert np.abs(got.cpu().numpy().astype(np.float64) - want).max() <= tol, key


@pytest.mark.parametrize("norm_type", [0, 1])
@pytest.mark.parametrize("B,per_row", [(1, 70001), (4097, 1), (37, 300), (5, 256), (3, 1000)])
def test_fisher_gradient_accumulation_paths(norm_type, B, per_row):
    rng = np.random.RandomState(B + per_row)
    A = np.concatenate([synth.fisher_A("diag531"), synth.fisher_A("tilted"), rng.randn(max(B - 2, 1), 3, 3)], 0)[:B] * (0.3 if norm_type == 0 else 1.0)
    n = B * per_row
    R = synth.uniform_rotations(n, seed=5)
    g = rng.randn(n) / np.sqrt(per_row)
    At = torch.from_numpy(A).requires_grad_(True)
    Rt = torch.from_numpy(R).double()
    lp64 = orc.fisher_log_prob(Rt, At, torch.float64, norm_type=norm_type)
    (lp64 * torch.from_numpy(g)).sum().backward()
    Ag = torch.from_numpy(A).float().cuda().requires_grad_(True)
    lp = MatrixFisherN(Ag, norm_type=norm_type)._log_prob(torch.from_numpy(R).float().cuda())
    (lp * torch.from_numpy(g).float().cuda()).sum().backward()
    assert np.abs(lp.detach().cpu().numpy() - lp64.detach().numpy()).max() < 2e-5 * max(1.0, float(lp64.detach().abs().max()))
    want = At.grad.numpy()
    assert np.abs(Ag.grad.cpu().numpy() - want).max() <= 3e-5 * max(1.0, np.abs(want).max())


def test_predicted_A_trains_through_the_flow_loss():
    """agent.py:55-65 shape of the computation: A comes out of a (here: linear) network per sample, the loss is the mean NLL of rotations
    under MatrixFisherN(A); the parameter gradient equals the oracle's."""
    torch.manual_seed(0)
    n = 192
    feat = torch.randn(n, 8)
    W = (0.4 * torch.randn(9, 8)).requires_grad_(True)
    bias = torch.tensor([3.0, 0, 0, 0, 2.0, 0, 0, 0, 1.0]).requires_grad_(True)
    R = torch.from_numpy(synth.uniform_rotations(n, seed=9))
    A64 = (feat.double() @ W.double().T + bias.double()).reshape(n, 3, 3)
    loss64 = -orc.fisher_log_prob(R.double(), A64, torch.float64).mean()
    gW64, gb64 = torch.autograd.grad(loss64, [W, bias])
    Wg = W.detach().cuda().requires_grad_(True)
    bg = bias.detach().cuda().requires_grad_(True)
    A = (feat.cuda() @ Wg.T + bg).reshape(n, 3, 3)
    loss = -MatrixFisherN(A)._log_prob(R.float().cuda()).mean()
    loss.backward()
    assert abs(float(loss) - float(loss64)) < 1e-5 * max(1.0, abs(float(loss64)))
    assert (Wg.grad.cpu() - gW64).abs().max() < 2e-5 * max(1.0, float(gW64.abs().max()))
    assert (bg.grad.cpu() - gb64).abs().max() < 2e-5 * max(1.0, float(gb64.abs().max()))


def test_monte_carlo_normaliser_norm_type_2():
    """norm_type 2 (utils/fisher.py:98-101): c = log mean_k exp(tr(R_k^T A)) over uniform rotations.  Checked against the same estimator on
    the oracle's side (torch uniform rotations from normalised Gaussian quaternions, 4e6 samples, fp64) within the Monte-Carlo error, and
    against an independent quadrature-free identity: for A = 0 the mean is exactly 1 (c = 0)."""
    A = synth.fisher_A("tilted")
    g = torch.Generator().manual_seed(5)
    q = torch.randn(4_000_000, 4, generator=g, dtype=torch.float64)
    q = q / q.norm(dim=-1, keepdim=True)
    Rk = orc.quaternion_to_matrix(q)
    tr = (Rk * torch.from_numpy(A).double()).sum((-1, -2))
    want = float(torch.logsumexp(tr, 0) - np.log(tr.numel()))
    rel_sd = float((tr - tr.max()).exp().std() / (tr - tr.max()).exp().mean() / np.sqrt(tr.numel()))
    torch.manual_seed(11)
    base = MatrixFisherN(torch.from_numpy(A).cuda(), norm_type=2, approx_num=4_000_000)
    got = float(base.log_const()[0])
    assert abs(got - want) < 6 * rel_sd + 1e-4, (got, want, rel_sd)
    R = torch.from_numpy(synth.uniform_rotations(256, seed=3)).cuda()
    lp = base._log_prob(R)
    tr_r = (R.cpu().double() * torch.from_numpy(A).double()).sum((-1, -2))
    assert (lp.cpu().double() - (tr_r - got)).abs().max() < 1e-4
    zero = MatrixFisherN(torch.zeros(1, 3, 3).cuda(), norm_type=2, approx_num=100_000)
    assert abs(float(zero.log_const()[0])) < 1e-5
    with pytest.raises(RuntimeError):
        MatrixFisherN(torch.zeros(2, 3, 3).cuda(), norm_type=2, approx_num=10)
    with pytest.raises(TypeError):
        MatrixFisherN(torch.zeros(1, 3, 3).cuda(), norm_type=2)
    with pytest.raises(NotImplementedError):
        MatrixFisherN(torch.zeros(1, 3, 3).cuda(), norm_type=3)
