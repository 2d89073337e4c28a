"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (torch, dtype-generic fp32/fp64) of the reference's SO(3) normalizing-flow density path.
Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may import this module; the shipped
package ``rotationnormflow_amd`` never does (its ops raise if the HIP library is missing).

Parity pinning: the reference holds no tests or golden vectors for this path, so the oracle is
pinned against outputs of the reference itself, generated in the build container by tests/golden/make_golden.py
(which imports /root/reference with the stand-ins under oracle/stubs/) and committed under tests/golden/*.npz;
tests/test_oracle_golden.py checks every fixture.  The one third-party boundary (pytorch3d 0.7.5 quaternion
conversions, source absent) is restated from its published semantics -- "parity unpinned" there, see DESIGN.md.

The formulas deliberately follow the reference step by step (3-D Moebius map, explicit [K,3,3] Jacobian, 15-step
bisection) rather than the closed forms the HIP kernels use, so that kernel-vs-oracle agreement is a real check.

All citations are relative to /root/reference/.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch

TWO_PI = 2.0 * math.pi

# flow/flow.py:13-15 -- (x-column, y-column, z-column) per exchange count mod 6
PERMUTE_ROWS = ((0, 1, 2), (1, 2, 0), (2, 0, 1), (0, 1, 2), (1, 2, 0), (2, 0, 1))


# --------------------------------------------------------------------------------------------------------------
# layer list (flow/flow.py:19-51, flow/mobiusflow.py:7-14, flow/affineflow.py:5-73 -- 16Trans family only)
# --------------------------------------------------------------------------------------------------------------
def _affine_kind(cfg, first_layer_condition=False):
    """Subset of get_affine (flow/affineflow.py:5-73): the 4x4 quaternion-affine family ('uncond16', 'cond16'), its
    unconditional LU parameterisation ('lu16') and the unconditional SVD rotation ('rot16')."""
    rot, lu = cfg.rot, bool(getattr(cfg, "lu", 0))
    if first_layer_condition and rot == "16UnTrans":           # affineflow.py:7-11
        return "clu16" if lu else "cond16"
    if first_layer_condition and rot == "16UnRot":             # affineflow.py:12-13
        return "crot16"
    if cfg.condition:
        if rot == "16Trans":                                    # affineflow.py:16-20
            return "clu16" if lu else "cond16"
        if rot == "16Rot":                                      # affineflow.py:41-42
            return "crot16"
        if rot == "9TransLSmith" and lu:                        # affineflow.py:34-36
            return "clu9"
        if rot == "16UnTrans":                                  # affineflow.py:21-25
            return "lu16" if lu else "uncond16"
        if rot == "16UnRot":                                    # affineflow.py:43-44
            return "rot16"
        if rot == "36Trans":                                    # affineflow.py:26-27
            return "cgs36"
        if rot == "9TransLSmith" and not lu:                    # affineflow.py:34-38
            return "cgs9"
        if rot == "9TransRSmith":                               # affineflow.py:39-40
            return "csmithr9"
        if rot == "9TransLSVD":                                 # affineflow.py:30-31
            return "csvdl9"
        if rot == "9TransRSVD":                                 # affineflow.py:32-33
            return "csvdr9"
    else:
        if rot == "16Trans":                                    # affineflow.py:50-54
            return "lu16" if lu else "uncond16"
        if rot == "16Rot":                                      # affineflow.py:70-71
            return "rot16"
        if rot == "36Trans":                                    # affineflow.py:55-56
            return "gs36"
        if rot == "9TransLSVD":                                 # affineflow.py:58-59
            return "svdl9"
        if rot == "9TransRSVD":                                 # affineflow.py:60-61
            return "svdr9"
        if rot == "9TransLSmith":                               # affineflow.py:62-66
            return "gs9lu" if lu else "gs9"
        if rot == "9TransRSmith":                               # affineflow.py:67-68
            return "smithr9"
    if rot in ("36Trans", "9TransLSVD", "9TransRSVD", "9TransLSmith", "9TransRSmith", "16Rot", "16UnRot"):
        raise NotImplementedError(f"oracle does not restate rot={rot!r} for condition={cfg.condition}")
    return None                                                 # affineflow.py:45-46,72-73


def feature_dim_of(cfg):
    """flow/flow.py:29-34."""
    if not cfg.condition:
        return 0
    fd = 32 if cfg.feature_dim is None else cfg.feature_dim
    if getattr(cfg, "embedding", 0):
        fd += cfg.embedding_dim
    return fd


def layer_kinds(cfg):
    """Ordered list of layer kinds ('mobius' | 'uncond16' | 'cond16'), index = position in Flow.layers."""
    kinds = []
    if cfg.last_affine:                                         # flow.py:37-39 (appended unguarded)
        k = _affine_kind(cfg, first_layer_condition=True)
        if k is None:
            raise TypeError("reference would append None here and fail at the first call (flow.py:38,65)")
        kinds.append(k)
    for i in range(cfg.layers):                                 # flow.py:41-48
        if getattr(cfg, "dist", "mobiusflow") != "noflow":
            kinds.append("mobius")
        k = _affine_kind(cfg)
        if k is not None and (i != cfg.layers - 1 or cfg.first_affine):
            kinds.append(k)
    return kinds


def state_shapes(cfg):
    """{state-dict key: shape} the reference's Flow(cfg).state_dict() holds."""
    fd = feature_dim_of(cfg)
    K = cfg.segments
    shapes = {}

    def mlp(prefix, ni, no):
        shapes[f"{prefix}.fc_first.weight"] = (64, ni)
        shapes[f"{prefix}.fc_first.bias"] = (64,)
        for j in (1, 3, 5):
            shapes[f"{prefix}.layers.{j}.weight"] = (64, 64)
            shapes[f"{prefix}.layers.{j}.bias"] = (64,)
        shapes[f"{prefix}.fc_last.weight"] = (no, 64)
        shapes[f"{prefix}.fc_last.bias"] = (no,)

    for i, kind in enumerate(layer_kinds(cfg)):
        if kind == "mobius":
            mlp(f"layers.{i}.conditioner", 3 + (fd if cfg.condition else 0), 4 * K)
        elif kind == "uncond16":
            shapes[f"layers.{i}.mat"] = (1, 4, 4)
        elif kind == "cond16":
            mlp(f"layers.{i}.net", fd, 16)
        elif kind in ("cgs9", "csmithr9", "csvdl9", "csvdr9"):  # squeezetrans.py:237, rottrans.py:111,141,171
            mlp(f"layers.{i}.net", fd, 9)
        elif kind == "cgs36":                                   # squeezetrans.py:337
            mlp(f"layers.{i}.net", fd, 36)
        elif kind == "crot16":                                  # ConditionRot, rottrans.py:40
            mlp(f"layers.{i}.net", fd, 16)
        elif kind in ("clu16", "clu9"):                         # ConditionLU(C, F), squeezetrans.py:94-119
            C = 4 if kind == "clu16" else 3
            for name, shp in (("w_p", (C, C)), ("u_mask", (C, C)), ("l_mask", (C, C)), ("s_sign", (C,)), ("l_eye", (C, C))):
                shapes[f"layers.{i}.net.{name}"] = shp
            mlp(f"layers.{i}.net.w_l_net", fd, C * C)
            mlp(f"layers.{i}.net.w_u_net", fd, C * C)
            mlp(f"layers.{i}.net.w_s_net", fd, C)
        elif kind == "lu16":                                    # UnconditionLU(4), squeezetrans.py:76-83
            for name, shp in (("w_p", (4, 4)), ("u_mask", (4, 4)), ("l_mask", (4, 4)), ("s_sign", (4,)), ("l_eye", (4, 4)),
                              ("w_l", (4, 4)), ("w_s", (4,)), ("w_u", (4, 4))):
                shapes[f"layers.{i}.mat.{name}"] = shp
        elif kind == "rot16":                                   # UnconditionRot, rottrans.py:11-12
            shapes[f"layers.{i}.rot"] = (1, 4, 4)
        elif kind in ("gs9", "svdl9", "svdr9", "smithr9"):     # squeezetrans.py:253, rottrans.py:97,127,157
            shapes[f"layers.{i}.mat"] = (3, 3)
        elif kind == "gs36":                                    # squeezetrans.py:353
            shapes[f"layers.{i}.mat"] = (6, 6)
        elif kind == "gs9lu":                                   # UnconditionLU(3), squeezetrans.py:267
            for name, shp in (("w_p", (3, 3)), ("u_mask", (3, 3)), ("l_mask", (3, 3)), ("s_sign", (3,)), ("l_eye", (3, 3)),
                              ("w_l", (3, 3)), ("w_s", (3,)), ("w_u", (3, 3))):
                shapes[f"layers.{i}.mat.{name}"] = shp
    return shapes


# --------------------------------------------------------------------------------------------------------------
# conditioner MLP (flow/condition.py:24-30)
# --------------------------------------------------------------------------------------------------------------
def conditioner(x, p, prefix):
    lin = torch.nn.functional.linear
    x0 = lin(x, p[f"{prefix}.fc_first.weight"], p[f"{prefix}.fc_first.bias"])
    h = x0
    for j in (1, 3, 5):                                         # ReLU, Linear pairs (condition.py:15-20)
        h = lin(torch.relu(h), p[f"{prefix}.layers.{j}.weight"], p[f"{prefix}.layers.{j}.bias"])
    h = torch.relu(x0 + h)                                      # condition.py:29
    return lin(h, p[f"{prefix}.fc_last.weight"], p[f"{prefix}.fc_last.bias"])


# --------------------------------------------------------------------------------------------------------------
# Moebius coupling layer (flow/mobiusflow.py)
# --------------------------------------------------------------------------------------------------------------
def _h(z, w):
    """mobiusflow.py:17-24.  z [N,3], w [N,K,3] -> [N,K,3]."""
    wn = torch.norm(w, dim=-1, keepdim=True)
    d = z[:, None, :] - w
    return (1 - wn ** 2) / (torch.norm(d, dim=-1, keepdim=True) ** 2) * d - w


def _wrapped_angle(hz, r, v):
    """atan2(h.v, h.r) mapped to [0, 2pi)  (mobiusflow.py:94-99, 234-239)."""
    ang = torch.atan2(torch.einsum("nki,ni->nk", hz, v), torch.einsum("nki,ni->nk", hz, r))
    return torch.where(ang >= 0, ang, ang + TWO_PI)


def _segment_params(cond_in, y, p, prefix, K):
    """conditioner output -> (normalised softplus weights [N,K], squashed projected w [N,K,3])
    (mobiusflow.py:57-63,69-72 and 140-147,152-155)."""
    c = conditioner(cond_in, p, prefix)
    sw, w = torch.split(c, [K, 3 * K], dim=1)
    w = w.reshape(-1, K, 3)
    eye = torch.eye(3, dtype=y.dtype)
    proj = eye[None] - torch.einsum("ni,nj->nij", y, y)
    w = torch.einsum("nij,nkj->nki", proj, w)
    sw = torch.nn.functional.softplus(sw)
    sw = sw / sw.sum(dim=-1, keepdim=True)
    w = 0.7 / (1 + torch.norm(w, dim=-1, keepdim=True)) * w
    return sw, w


def _mobius_core(x, r, v, sw, w):
    """mobiusflow.py:90-125: transformed column and log-det via the explicit dh/dz Jacobian."""
    hz = _h(x, w)
    ang = (sw * _wrapped_angle(hz, r, v)).sum(dim=1, keepdim=True)
    tx = r * torch.cos(ang) + v * torch.sin(ang)

    zw = x[:, None, :] - w
    zwn = torch.norm(zw, dim=-1)
    zwu = zw / zwn[..., None]
    theta = torch.atan2((x * v).sum(-1), (x * r).sum(-1)).reshape(-1, 1)
    dz = -torch.sin(theta) * r + torch.cos(theta) * v
    eye = torch.eye(3, dtype=x.dtype)
    dh_dz = ((1 - torch.norm(w, dim=-1) ** 2)[..., None, None]
             * (eye[None, None] - 2 * torch.einsum("nki,nkj->nkij", zwu, zwu))
             / (zwn[..., None, None] ** 2))
    dh = torch.einsum("nkpq,nq->nkp", dh_dz, dz)
    return tx, torch.log((torch.norm(dh, dim=-1) * sw).sum(dim=1))


def _cross(a, b):
    return torch.linalg.cross(a, b, dim=-1)     # the reference's torch.cross(a, b) picks dim -1 for N != 3


def _unit(a):
    return a / a.norm(dim=-1, keepdim=True)


def _assemble(cx, cy, perm):
    """Third column by the branch at mobiusflow.py:75-79 / 172-176 and the column scatter at :80-83."""
    p0, p1, p2 = perm
    cz = _cross(cx, cy) if (p1 - p0) in (1, -2) else _cross(cy, cx)
    cz = _unit(cz)
    out = torch.empty(cx.shape[0], 3, 3, dtype=cx.dtype)
    out[..., p0] = cx
    out[..., p1] = cy
    out[..., p2] = cz
    return out


def mobius_forward(R, perm, feature, p, prefix, K):
    """mobiusflow.py:46-85."""
    x = R[..., perm[0]]
    y = R[..., perm[1]]
    cond_in = y if feature is None else torch.cat((y, feature), dim=-1)
    sw, w = _segment_params(cond_in, y, p, prefix, K)
    r = _unit(-x)
    v = _unit(_cross(y, r))
    tx, ldj = _mobius_core(x, r, v, sw, w)
    return _assemble(tx, y, perm), ldj


def _theta_map(theta, r, v, sw, w):
    """BinFind._forward_theta (mobiusflow.py:226-245)."""
    z = r * torch.cos(theta) + v * torch.sin(theta)
    return (sw * _wrapped_angle(_h(z, w), r, v)).sum(dim=1, keepdim=True)


def _bisect(target, r, v, sw, w):
    """BinFind.forward (mobiusflow.py:189-224): batch-global stop test, returns the LAST midpoint."""
    a = torch.full_like(target, math.pi / 2)
    b = torch.full_like(target, 3 * math.pi / 2)
    mid = None
    it = 1
    while abs(torch.max(b - a)) >= 1e-4:
        mid = (a + b) / 2
        f = _theta_map(mid, r, v, sw, w) - target
        if it > 100:                                            # mobiusflow.py:212-214
            break
        a = a + (b - a) / 2 * (f < 0)
        b = b - (b - a) / 2 * (f >= 0)
        it += 1
    return mid


class _BinFind(torch.autograd.Function):
    """BinFind (mobiusflow.py:186-273) with its custom backward: the implicit-function gradient of the root of
    _forward_theta(theta; r, v, weights, w) = y, evaluated at the bisection's returned midpoint --
    d theta / dy = 1 / F_theta and d theta / dp = -F_p / F_theta for p in (r, v, weights, w), zero where F_theta == 0 (:262-272)."""

    @staticmethod
    def forward(ctx, y, r, v, sw, w):
        theta = _bisect(y.detach(), r.detach(), v.detach(), sw.detach(), w.detach())
        ctx.save_for_backward(theta, r.detach(), v.detach(), sw.detach(), w.detach())
        return theta.clone()

    @staticmethod
    def backward(ctx, g):
        theta, r, v, sw, w = (t.clone().requires_grad_(True) for t in ctx.saved_tensors)
        with torch.enable_grad():
            gt, gr, gv, gsw, gw = torch.autograd.grad(_theta_map(theta, r, v, sw, w), (theta, r, v, sw, w), torch.ones_like(g))
        ok = gt != 0
        inv = torch.where(ok, 1 / gt, torch.zeros_like(gt))
        return inv * g, -gr * inv * g, -gv * inv * g, -gsw * inv * g, -gw * inv[..., None] * g[..., None]


def mobius_inverse(R, perm, feature, p, prefix, K):
    """mobiusflow.py:127-183."""
    tx = R[..., perm[0]]
    ty = R[..., perm[1]]
    cond_in = ty if feature is None else torch.cat((ty, feature), dim=-1)
    sw, w = _segment_params(cond_in, ty, p, prefix, K)
    r = _unit(-tx)
    v = _unit(_cross(ty, r))
    tt = torch.atan2((tx * v).sum(-1), (tx * r).sum(-1)).reshape(-1, 1)
    tt = torch.where(tt >= 0, tt, tt + TWO_PI)
    tt = torch.where((tt - TWO_PI).abs() < 1e-4, torch.zeros_like(tt), tt)
    theta = _BinFind.apply(tt, r, v, sw, w) if torch.is_grad_enabled() else _bisect(tt, r, v, sw, w)
    x = r * torch.cos(theta) + v * torch.sin(theta)
    _, ldj = _mobius_core(x, r, v, sw, w)
    return _assemble(x, ty, perm), -ldj


# --------------------------------------------------------------------------------------------------------------
# quaternion affine layer (flow/squeezetrans.py:10-55,161-174) + the pytorch3d conversions it relies on
# --------------------------------------------------------------------------------------------------------------
def matrix_to_quaternion(M):
    """Published pytorch3d 0.7.5 rule (source absent): four candidates, pick the largest |q_i|, 0.1 floor."""
    m = M.reshape(-1, 9)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.unbind(-1)
    qa = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], -1)
    qa = torch.sqrt(torch.clamp(qa, min=0))
    cand = torch.stack([
        torch.stack([qa[:, 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
        torch.stack([m21 - m12, qa[:, 1] ** 2, m10 + m01, m02 + m20], -1),
        torch.stack([m02 - m20, m10 + m01, qa[:, 2] ** 2, m12 + m21], -1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, qa[:, 3] ** 2], -1),
    ], -2)
    cand = cand / (2 * torch.clamp(qa, min=0.1))[..., None]
    idx = qa.argmax(-1)
    return cand[torch.arange(m.shape[0]), idx]


def quaternion_to_matrix(q):
    w, x, y, z = q.unbind(-1)
    s2 = 2.0 / (q * q).sum(-1)
    return torch.stack([
        1 - s2 * (y * y + z * z), s2 * (x * y - z * w), s2 * (x * z + y * w),
        s2 * (x * y + z * w), 1 - s2 * (x * x + z * z), s2 * (y * z - x * w),
        s2 * (x * z - y * w), s2 * (y * z + x * w), 1 - s2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


def _det3(A):
    """squeezetrans.py:10-14."""
    c0 = A[..., 1, 1] * A[..., 2, 2] - A[..., 1, 2] * A[..., 2, 1]
    c1 = A[..., 1, 2] * A[..., 2, 0] - A[..., 1, 0] * A[..., 2, 2]
    c2 = A[..., 1, 0] * A[..., 2, 1] - A[..., 1, 1] * A[..., 2, 0]
    return c0 * A[..., 0, 0] + c1 * A[..., 0, 1] + c2 * A[..., 0, 2]


def _det4(A):
    """squeezetrans.py:17-22 (cofactor expansion along row 0)."""
    sub = A[..., 1:, :]
    t = [A[..., 0, j] * _det3(sub[..., [c for c in range(4) if c != j]]) for j in range(4)]
    return t[0] - t[1] + t[2] - t[3]


def affine16(M, R):
    """calculate_16 (squeezetrans.py:33-38).  M [1,4,4] or [N,4,4]."""
    q = matrix_to_quaternion(R)
    q = M @ q.reshape(-1, 4, 1)
    ln = q.norm(dim=-2, keepdim=True)
    Rt = quaternion_to_matrix((q / ln).reshape(-1, 4))
    return Rt, _det4(M).abs().log() - 4 * ln.reshape(-1).log()


def lu16_matrix(p, prefix):
    """UnconditionLU.forward (squeezetrans.py:85-91): P (L*mask + I) (U*mask + diag(sign * exp(s)))."""
    g = lambda n: p[f"{prefix}.{n}"]  # noqa: E731
    return (g("w_p") @ (g("w_l") * g("l_mask") + g("l_eye"))
            @ ((g("w_u") * g("u_mask")) + torch.diag(g("s_sign") * torch.exp(g("w_s"))))).unsqueeze(0)


def rot16_matrix(p, key):
    """UnconditionRot (rottrans.py:15-17): U^T V from the SVD of the 4x4 parameter."""
    U, S, Vh = torch.linalg.svd(p[key])
    return U.transpose(-1, -2) @ Vh.transpose(-1, -2)


def rot16_apply(M, R):
    """rottrans.py:18-23: rotate the quaternion, no renormalisation, log-det exactly 0."""
    q = matrix_to_quaternion(R)
    q = (M @ q.reshape(-1, 4, 1)).reshape(-1, 4)
    return quaternion_to_matrix(q), torch.zeros(R.shape[0], dtype=R.dtype)


# --------------------------------------------------------------------------------------------------------------
# 3x3 / 6x6 ablation layers (flow/squeezetrans.py:176-361, flow/rottrans.py:72-165), unconditional variants
# --------------------------------------------------------------------------------------------------------------
# generators of the tangent directions the reference differentiates along (squeezetrans.py:200-201, 295-296)
_GEN = torch.tensor([[[0., 1, 0], [-1, 0, 0], [0, 0, 0]], [[0, 0, 1], [0, 0, 0], [-1, 0, 0]], [[0, 0, 0], [0, 0, 1], [0, -1, 0]]])


def _gram_schmidt_tangent(a0, a1, da0, da1):
    """Gram-Schmidt of the two columns (a0, a1) [N,3] to a rotation, carried together with three tangent derivatives
    (da0, da1) [3,N,3] (forward mode; accp_normalize / accp_cross / the body of calculate_9, squeezetrans.py:176-231).
    Returns (R' [N,3,3], log|det of the 3x3 matrix of tangent images|)."""
    def normalize(a, da):
        n = a.norm(dim=-1, keepdim=True)
        t = a / n
        return t, (da - t * (t * da).sum(-1, keepdim=True)) / n
    t0, dt0 = normalize(a0, da0)
    dot = (t0 * a1).sum(-1, keepdim=True)
    ddot = (dt0 * a1).sum(-1, keepdim=True) + (t0 * da1).sum(-1, keepdim=True)
    t1, dt1 = normalize(a1 - dot * t0, da1 - ddot * t0 - dot * dt0)
    t2 = torch.linalg.cross(t0, t1)
    dt2 = torch.linalg.cross(dt0, t1.expand_as(dt0)) + torch.linalg.cross(t0.expand_as(dt1), dt1)
    Rt = torch.stack([t0, t1, t2], dim=-1)                               # columns
    dRt = torch.stack([dt0, dt1, dt2], dim=-1)                            # [3,N,3,3]
    delta = dRt @ Rt.transpose(-1, -2)                                    # skew: dR' R'^T
    vec = torch.stack([delta[..., 0, 1], delta[..., 0, 2], delta[..., 1, 2]], dim=-1)   # [3,N,3]
    return Rt, _det3(vec.transpose(0, 1)).abs().log()


def gs9(M, R):
    """calculate_9 (squeezetrans.py:199-231): Gram-Schmidt of M R, tangent directions R G_k (right multiplication)."""
    G = _GEN.to(R.dtype)
    MR = M.reshape(-1, 3, 3) @ R
    dMR = torch.einsum("nab,kbc->knac", MR, G)
    return _gram_schmidt_tangent(MR[..., 0], MR[..., 1], dMR[..., 0], dMR[..., 1])


def gs36(M, R):
    """calculate_36 (squeezetrans.py:293-331): the first two columns of R as a 6-vector times M [6,6], Gram-Schmidt; tangent
    directions G_k R (left multiplication)."""
    G = _GEN.to(R.dtype)
    M = M.reshape(-1, 6, 6)
    dR = torch.einsum("kab,nbc->knac", G, R)
    v = torch.cat([R[..., 0], R[..., 1]], dim=-1)
    dv = torch.cat([dR[..., 0], dR[..., 1]], dim=-1)
    Mn = M.expand(R.shape[0], 6, 6)                                     # one shared matrix or one per sample
    tv = torch.einsum("nab,nb->na", Mn, v)
    dtv = torch.einsum("nab,knb->kna", Mn, dv)
    return _gram_schmidt_tangent(tv[..., :3], tv[..., 3:], dtv[..., :3], dtv[..., 3:])


def lu_matrix(p, prefix):
    """UnconditionLU.forward without the batch axis (squeezetrans.py:85-91)."""
    g = lambda n: p[f"{prefix}.{n}"]  # noqa: E731
    return g("w_p") @ (g("w_l") * g("l_mask") + g("l_eye")) @ ((g("w_u") * g("u_mask")) + torch.diag(g("s_sign") * torch.exp(g("w_s"))))


def _polar_rotation(X):
    """U V^T of the SVD X = U S V^T (rottrans.py:73-76, 79-82)."""
    U, S, Vh = torch.linalg.svd(X)
    return U @ Vh


def svdl9(M, R):
    """calculate_9_l (rottrans.py:72-76): nearest orthogonal matrix to M R; log-det reported as 0."""
    return _polar_rotation(M.reshape(-1, 3, 3) @ R), torch.zeros(R.shape[0], dtype=R.dtype)


def svdr9(M, R):
    """calculate_9_r (rottrans.py:79-82): nearest orthogonal matrix to R M."""
    return _polar_rotation(R @ M.reshape(-1, 3, 3)), torch.zeros(R.shape[0], dtype=R.dtype)


def smithr9(M, R, inverse=False):
    """calculate_9_r_smith (rottrans.py:85-96): R times the Gram-Schmidt rotation of the columns of M (its transpose for the inverse).
    M [3,3] or [N,3,3]."""
    m0 = M[..., :, 0] / M[..., :, 0].norm(dim=-1, keepdim=True)
    m1 = M[..., :, 1] - (m0 * M[..., :, 1]).sum(-1, keepdim=True) * m0
    m1 = m1 / m1.norm(dim=-1, keepdim=True)
    Q = torch.stack([m0, m1, torch.linalg.cross(m0, m1)], dim=-1)
    if inverse:
        Q = Q.transpose(-1, -2)
    return R @ Q, torch.zeros(R.shape[0], dtype=R.dtype)


def cond9_matrix(feature, p, prefix):
    """Condition9Trans / Condition9Rot* (squeezetrans.py:240-241, rottrans.py:114-115): I + reshape(net(feature), 3, 3)."""
    return conditioner(feature, p, prefix).reshape(-1, 3, 3) + torch.eye(3, dtype=feature.dtype)[None]


def cond36_matrix(feature, p, prefix):
    """Condition36Trans (squeezetrans.py:339-341): I + reshape(net(feature), 6, 6)."""
    return conditioner(feature, p, prefix).reshape(-1, 6, 6) + torch.eye(6, dtype=feature.dtype)[None]


def cond_lu_matrix(feature, p, prefix, C):
    """ConditionLU.forward (squeezetrans.py:121-131).  NOTE torch.diag of the 2-D [N, C] tensor s_sign * exp(w_s_net(feature)) takes the
    diagonal ACROSS THE BATCH (entry i of sample i, i < C), and the resulting C-vector is broadcast onto every row of every sample's
    upper factor: the layer's output for one sample depends on the first C samples of the batch it travels in.  Restated as written."""
    wl = conditioner(feature, p, f"{prefix}.w_l_net").reshape(-1, C, C)
    wu = conditioner(feature, p, f"{prefix}.w_u_net").reshape(-1, C, C)
    ws = conditioner(feature, p, f"{prefix}.w_s_net")
    return torch.einsum("ab,nbc,ncd->nad", p[f"{prefix}.w_p"], wl * p[f"{prefix}.l_mask"] + p[f"{prefix}.l_eye"],
                        wu * p[f"{prefix}.u_mask"] + torch.diag(p[f"{prefix}.s_sign"] * torch.exp(ws)))


def cond_rot16_matrix(feature, p, prefix, inverse=False):
    """ConditionRot (rottrans.py:42-46, 55-59): U^T V of the batched SVD of I + reshape(net(feature), 4, 4); transposed for the inverse."""
    mat = conditioner(feature, p, prefix).reshape(-1, 4, 4) + torch.eye(4, dtype=feature.dtype)[None]
    U, S, V = torch.svd(mat)
    rot = U.transpose(-1, -2) @ V
    return rot.transpose(-1, -2) if inverse else rot


def cond16_matrix(feature, p, prefix):
    """Condition16Trans (squeezetrans.py:47-48)."""
    return conditioner(feature, p, prefix).reshape(-1, 4, 4) + torch.eye(4, dtype=feature.dtype)[None]


# --------------------------------------------------------------------------------------------------------------
# the flow stack (flow/flow.py:53-92)
# --------------------------------------------------------------------------------------------------------------
def _as_params(params, dtype):
    return {k: torch.as_tensor(v).to(dtype) for k, v in params.items()}


def flow_forward(cfg, params, R, feature=None, dtype=torch.float32, grad=False, rot_override=None):
    """Flow.forward (flow.py:53-72): returns (R' [N,3,3], ldj [N]).  grad=True keeps the autograd graph (params / R / feature
    given as torch tensors that require grad): the oracle for the training path's gradients (agent.py:79-80).
    rot_override {layer index: [N,4,4]}: a ConditionRot layer applies THESE orthogonal matrices instead of U^T V of its own SVD -- U^T V is
    defined by the SVD routine's sign choices, so the per-sample parity test hands the oracle the factors of the routine under test."""
    p = _as_params(params, dtype)
    R = torch.as_tensor(R).to(dtype)
    feature = None if (feature is None or not cfg.condition) else torch.as_tensor(feature).to(dtype)
    kinds = layer_kinds(cfg)
    K = cfg.segments
    ldj = torch.zeros(R.shape[0], dtype=dtype)
    count = 0
    with torch.set_grad_enabled(grad):
        for i, kind in enumerate(kinds):
            perm = PERMUTE_ROWS[count % 6]
            if kind == "mobius":
                R, l = mobius_forward(R, perm, feature, p, f"layers.{i}.conditioner", K)
            elif kind == "uncond16":
                R, l = affine16(p[f"layers.{i}.mat"], R)
            elif kind == "lu16":
                R, l = affine16(lu16_matrix(p, f"layers.{i}.mat"), R)                   # squeezetrans.py:151-152
            elif kind == "rot16":
                R, l = rot16_apply(rot16_matrix(p, f"layers.{i}.rot"), R)
            elif kind == "gs9":
                R, l = gs9(p[f"layers.{i}.mat"], R)                                     # squeezetrans.py:255-257
            elif kind == "gs9lu":
                R, l = gs9(lu_matrix(p, f"layers.{i}.mat"), R)                          # squeezetrans.py:269-271
            elif kind == "gs36":
                R, l = gs36(p[f"layers.{i}.mat"], R)                                    # squeezetrans.py:355-357
            elif kind == "svdl9":
                R, l = svdl9(p[f"layers.{i}.mat"], R)                                   # rottrans.py:99-101
            elif kind == "svdr9":
                R, l = svdr9(p[f"layers.{i}.mat"], R)                                   # rottrans.py:129-131
            elif kind == "smithr9":
                R, l = smithr9(p[f"layers.{i}.mat"], R)                                 # rottrans.py:159-161
            elif kind == "clu16":
                R, l = affine16(cond_lu_matrix(feature, p, f"layers.{i}.net", 4), R)    # squeezetrans.py:139-140
            elif kind == "clu9":
                R, l = gs9(cond_lu_matrix(feature, p, f"layers.{i}.net", 3) + torch.eye(3, dtype=R.dtype)[None], R)   # squeezetrans.py:269-271
            elif kind == "crot16":
                rot = torch.as_tensor(rot_override[i]).to(dtype) if rot_override and i in rot_override else cond_rot16_matrix(feature, p, f"layers.{i}.net")
                R, l = rot16_apply(rot, R)                                              # rottrans.py:42-53
            elif kind == "cgs9":
                R, l = gs9(cond9_matrix(feature, p, f"layers.{i}.net"), R)              # squeezetrans.py:239-242
            elif kind == "cgs36":
                R, l = gs36(cond36_matrix(feature, p, f"layers.{i}.net"), R)            # squeezetrans.py:339-342
            elif kind == "csmithr9":
                R, l = smithr9(cond9_matrix(feature, p, f"layers.{i}.net"), R)          # rottrans.py:173-176
            elif kind == "csvdl9":
                R, l = svdl9(cond9_matrix(feature, p, f"layers.{i}.net"), R)            # rottrans.py:113-116
            elif kind == "csvdr9":
                R, l = svdr9(cond9_matrix(feature, p, f"layers.{i}.net"), R)            # rottrans.py:143-146
            else:
                R, l = affine16(cond16_matrix(feature, p, f"layers.{i}.net"), R)
            ldj = ldj + l
            if kind == "mobius" or cfg.frequent_permute:        # flow.py:69-70
                count += 1
    return R, ldj


def flow_inverse(cfg, params, R, feature=None, dtype=torch.float32, grad=False, rot_override=None):
    """Flow.inverse (flow.py:74-92): returns (R [N,3,3], ldj_of_inverse_map [N]).  grad=True keeps the autograd graph (through
    BinFind's custom backward): the oracle for gradients through the inverse pass."""
    p = _as_params(params, dtype)
    R = torch.as_tensor(R).to(dtype)
    feature = None if (feature is None or not cfg.condition) else torch.as_tensor(feature).to(dtype)
    kinds = layer_kinds(cfg)
    K = cfg.segments
    ldj = torch.zeros(R.shape[0], dtype=dtype)
    count = len(kinds) if cfg.frequent_permute else cfg.layers  # flow.py:78-79
    with torch.set_grad_enabled(grad):
        for i in reversed(range(len(kinds))):
            kind = kinds[i]
            if kind == "mobius" or cfg.frequent_permute:        # flow.py:85-86 (decrement BEFORE use)
                count -= 1
            perm = PERMUTE_ROWS[count % 6]
            if kind == "mobius":
                R, l = mobius_inverse(R, perm, feature, p, f"layers.{i}.conditioner", K)
            elif kind == "uncond16":
                R, l = affine16(torch.linalg.inv(p[f"layers.{i}.mat"]), R)       # squeezetrans.py:171-174
            elif kind == "lu16":
                R, l = affine16(torch.linalg.inv(lu16_matrix(p, f"layers.{i}.mat")), R)   # squeezetrans.py:154-157
            elif kind == "rot16":
                R, l = rot16_apply(rot16_matrix(p, f"layers.{i}.rot").transpose(-1, -2), R)   # rottrans.py:26-28
            elif kind == "gs9":
                R, l = gs9(torch.linalg.inv(p[f"layers.{i}.mat"]), R)                   # squeezetrans.py:259-261
            elif kind == "gs9lu":
                R, l = gs9(torch.linalg.inv(lu_matrix(p, f"layers.{i}.mat")), R)        # squeezetrans.py:273-275
            elif kind == "gs36":
                R, l = gs36(torch.linalg.inv(p[f"layers.{i}.mat"]), R)                  # squeezetrans.py:359-361
            elif kind == "svdl9":
                R, l = svdl9(p[f"layers.{i}.mat"].transpose(-1, -2), R)                 # rottrans.py:103-105
            elif kind == "svdr9":
                R, l = svdr9(p[f"layers.{i}.mat"].transpose(-1, -2), R)                 # rottrans.py:133-135
            elif kind == "smithr9":
                R, l = smithr9(p[f"layers.{i}.mat"], R, inverse=True)                   # rottrans.py:163-165
            elif kind == "clu16":
                R, l = affine16(torch.linalg.inv(cond_lu_matrix(feature, p, f"layers.{i}.net", 4)), R)                  # squeezetrans.py:142-144
            elif kind == "clu9":
                R, l = gs9(torch.linalg.inv(cond_lu_matrix(feature, p, f"layers.{i}.net", 3) + torch.eye(3, dtype=R.dtype)[None]), R)   # :273-277
            elif kind == "crot16":
                rot = (torch.as_tensor(rot_override[i]).to(dtype).transpose(-1, -2) if rot_override and i in rot_override      # (forward orientation handed in)
                       else cond_rot16_matrix(feature, p, f"layers.{i}.net", inverse=True))
                R, l = rot16_apply(rot, R)                                              # rottrans.py:55-66
            elif kind == "cgs9":
                R, l = gs9(torch.linalg.inv(cond9_matrix(feature, p, f"layers.{i}.net")), R)        # squeezetrans.py:244-247
            elif kind == "cgs36":
                R, l = gs36(torch.linalg.inv(cond36_matrix(feature, p, f"layers.{i}.net")), R)      # squeezetrans.py:344-347
            elif kind == "csmithr9":
                R, l = smithr9(cond9_matrix(feature, p, f"layers.{i}.net"), R, inverse=True)        # rottrans.py:178-181
            elif kind == "csvdl9":
                R, l = svdl9(cond9_matrix(feature, p, f"layers.{i}.net").transpose(-1, -2), R)      # rottrans.py:118-121
            elif kind == "csvdr9":
                R, l = svdr9(cond9_matrix(feature, p, f"layers.{i}.net").transpose(-1, -2), R)      # rottrans.py:148-151
            else:
                R, l = affine16(torch.linalg.inv(cond16_matrix(feature, p, f"layers.{i}.net")), R)  # :51-55
            ldj = ldj + l
    return R, ldj


# --------------------------------------------------------------------------------------------------------------
# matrix-Fisher base log-density (utils/fisher.py:67-76,93-97,209-232)
# --------------------------------------------------------------------------------------------------------------
def proper_singular_values(A):
    """proper_svd_N (fisher.py:67-76): singular values with the last one sign-flipped by det(U)det(V)."""
    U, S, Vh = torch.linalg.svd(A)
    return torch.stack([S[:, 0], S[:, 1], S[:, 2] * torch.det(U) * torch.det(Vh)], dim=-1)   # out of place: differentiable w.r.t. A


def fisher_norm(S, norm_type=1):
    """matrix_fisher_norm_N (fisher.py:79-115) on the proper singular values S [B,3].  norm_type 0 follows the reference to the
    letter: ``(S**2).sum()`` has no ``dim`` (fisher.py:91), so the quadratic term is summed over the WHOLE batch of matrices.
    Types 2 (Monte-Carlo over pytorch3d random rotations) and 3 (an ODE on S[0], S[1], S[2], which index ROWS of the [B,3] tensor
    and fail for B < 3) are not restated."""
    if norm_type == 0:
        norm = 1.0 + 1.0 / 6.0 * (S ** 2).sum() + 1.0 / 6.0 * S[:, 0] * S[:, 1] * S[:, 2]
        return norm / torch.sum(S, dim=-1).exp()
    if norm_type == 1:
        return 1.0 / torch.sqrt(8 * math.pi * (S[:, 0] + S[:, 1]) * (S[:, 2] + S[:, 1]) * (S[:, 0] + S[:, 2]))
    raise NotImplementedError(f"norm_type={norm_type}")


def fisher_log_prob(R, A, dtype=torch.float32, norm_type=1):
    """MatrixFisherN(A, norm_type)._log_prob(R) (fisher.py:79-97,217-232).  A [B,3,3] broadcasts over N/B consecutive samples
    (fisher.py:226).  Built from differentiable torch ops: autograd of this function w.r.t. A is the checker of the d/dA kernel."""
    A = torch.as_tensor(A).to(dtype)
    R = torch.as_tensor(R).to(dtype).reshape(A.shape[0], -1, 3, 3)
    S = proper_singular_values(A)
    norm = fisher_norm(S, norm_type)
    tr = (R * A.reshape(-1, 1, 3, 3)).sum(-1).sum(-1)
    return ((tr - S.sum(-1).reshape(-1, 1)) - norm.log().reshape(-1, 1)).reshape(-1)


# --------------------------------------------------------------------------------------------------------------
# matrix-Fisher sampler (utils/fisher.py:14-64,117-207,234-243): Bingham on S^3 by ACG-envelope rejection
# --------------------------------------------------------------------------------------------------------------
def proper_svd(A):
    """proper_svd (fisher.py:53-64) for one 3x3 matrix: (U, S, V) with U[:,2], S[2], V[:,2] sign-fixed so det U = det V = +1."""
    U, S, Vh = torch.linalg.svd(A)
    V = Vh.transpose(-1, -2)
    dU, dV = torch.det(U), torch.det(V)
    U = U.clone(); S = S.clone(); V = V.clone()
    U[:, 2] *= dU
    S[2] *= dU * dV
    V[:, 2] *= dV
    return U, S, V


def bingham_constants(S, b=1.5):
    """fisher.py:183-193: diagonal Bingham parameter, ACG envelope Omega, Gaussian std, rejection bound M*."""
    lam = torch.zeros(4, dtype=S.dtype)
    lam[1] = 2 * (S[1] + S[2])
    lam[2] = 2 * (S[0] + S[2])
    lam[3] = 2 * (S[0] + S[1])
    omega = torch.ones(4, dtype=S.dtype) + 2 * lam / b
    return lam, omega, omega ** (-0.5), math.exp(-(4 - b) / 2) * ((4 / b) ** 2)


def sample_matrix_fisher(A, num_samples, b=1.5, oversampling_ratio=8):
    """sample_matrix_fisher + sample_bingham + quat_to_rotmat (fisher.py:14-50,117-207) for one A [3,3], torch global RNG
    consumed in the reference's order (randn(n*8, 4) then rand(n*8) per attempt)."""
    U, S, V = proper_svd(A)
    lam, omega, std, m_star = bingham_constants(S, b)
    while True:
        eps = torch.randn(num_samples * oversampling_ratio, 4).float()
        y = std * eps
        x = y / torch.norm(y, dim=1, keepdim=True)
        p_bing = torch.exp(-torch.einsum("bn,n,bn->b", x, lam, x))
        p_acg = torch.einsum("bn,n,bn->b", x, omega, x) ** (-2)
        w = torch.rand(num_samples * oversampling_ratio)
        acc = w < p_bing / (m_star * p_acg)
        if int(acc.sum()) >= num_samples:
            q = x[acc, :][:num_samples, :]
            break
    q = q / q.norm(p=2, dim=1, keepdim=True)
    Rq = quaternion_to_matrix(q)                     # fisher.py:14-50 is the same real-first formula for unit q
    return U @ Rq.to(U.dtype) @ V.T


def fisher_sample(A, num_samples):
    """MatrixFisherN._sample (fisher.py:234-243): [B, num_samples, 3, 3], rows of A sampled one after the other."""
    A = torch.as_tensor(A)
    return torch.stack([sample_matrix_fisher(A[i], num_samples) for i in range(A.shape[0])])


# --------------------------------------------------------------------------------------------------------------
# NLL accumulation (agent.py:55-65,226-229; eval_uncondition.py:43-45)
# --------------------------------------------------------------------------------------------------------------
def log_prob(cfg, params, R, feature=None, A=None, dtype=torch.float32):
    """per-sample log p(R) = ldj_total + base(R')  (uniform base => 0) and the mean NLL."""
    Rt, ldj = flow_forward(cfg, params, R, feature, dtype)
    lp = ldj if A is None else ldj + fisher_log_prob(Rt, A, dtype)
    return lp, float(-(lp.double().mean()))


def make_config(**kw):
    """Attribute bag with the fields Flow/get_mobius/get_affine read (SURVEY 8(b) 'constructor')."""
    base = dict(layers=24, segments=64, condition=0, feature_dim=None, embedding=0, embedding_dim=0,
                last_affine=0, first_affine=1, frequent_permute=0, dist="mobiusflow", rot="16Trans", lu=0)
    base.update(kw)
    return SimpleNamespace(**base)
