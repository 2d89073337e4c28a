"""TEST INFRASTRUCTURE ONLY -- restatement of the published semantics of four pytorch3d.transforms functions.

pytorch3d (pinned 0.7.5, reference README.md:29) is a third-party dependency that is NOT vendored under
/root/reference and is not installed in this image.  The reference's hot path calls
    matrix_to_quaternion / quaternion_to_matrix   (flow/squeezetrans.py:34,37; flow/rottrans.py:18,20; utils/fisher.py:222,243)
    random_rotations / random_rotation            (utils/fisher.py:99; utils/sd.py:23)
This file restates their documented behaviour (real-part-first quaternions; the 4-candidate, arg-max-denominator
matrix->quaternion rule with the 0.1 floor; the un-normalised-quaternion-safe quaternion->matrix formula;
normalised-Gaussian random quaternions with the sign fixed by copysign) so that the reference can be imported in
this container to generate golden vectors.  "Parity unpinned" at this boundary: none of the reference's own tests
pins these functions; see DESIGN.md (oracle section) for why results are insensitive to the candidate chosen.
"""
import torch


def _sqrt_clamped(x):
    # sqrt(max(0, x)) with a zero (sub)gradient at 0
    out = torch.zeros_like(x)
    pos = x > 0
    out[pos] = torch.sqrt(x[pos])
    return out


def quaternion_to_matrix(quaternions):
    w, x, y, z = torch.unbind(quaternions, -1)
    s2 = 2.0 / (quaternions * quaternions).sum(-1)
    rows = (
        1 - s2 * (y * y + z * z), s2 * (x * y - z * w), s2 * (x * z + y * w),
        s2 * (x * y + z * w), 1 - s2 * (x * x + z * z), s2 * (y * z - x * w),
        s2 * (x * z - y * w), s2 * (y * z + x * w), 1 - s2 * (x * x + y * y),
    )
    return torch.stack(rows, -1).reshape(quaternions.shape[:-1] + (3, 3))


def matrix_to_quaternion(matrix):
    lead = matrix.shape[:-2]
    m = matrix.reshape(lead + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(m, -1)
    q_abs = _sqrt_clamped(torch.stack([
        1.0 + m00 + m11 + m22,
        1.0 + m00 - m11 - m22,
        1.0 - m00 + m11 - m22,
        1.0 - m00 - m11 + m22,
    ], dim=-1))
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1),
    ], dim=-2)
    floor = torch.tensor(0.1, dtype=q_abs.dtype, device=q_abs.device)
    cand = cand / (2.0 * q_abs[..., None].max(floor))
    pick = torch.nn.functional.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5
    return cand[pick, :].reshape(lead + (4,))


def random_quaternions(n, dtype=None, device=None):
    o = torch.randn((n, 4), dtype=dtype, device=device)
    s = (o * o).sum(1)
    return o / torch.copysign(torch.sqrt(s), o[:, 0])[:, None]


def random_rotations(n, dtype=None, device=None):
    return quaternion_to_matrix(random_quaternions(n, dtype=dtype, device=device))


def random_rotation(dtype=None, device=None):
    return random_rotations(1, dtype, device)[0]
