"""Mirror of the reference's flow/rottrans.py names (SVD / Smith rotation layers, ldj = 0).  None of the BASELINE
configurations uses them; they are declared for the registry and fail loudly until their kernels are built."""
import torch
import torch.nn as nn

from .. import runtime
from .mobiusflow import _SingleLayer
from .squeezetrans import _not_built


class UnconditionRot(nn.Module, _SingleLayer):
    """Rotate the quaternion by the orthogonal 4x4 matrix U^T V of the SVD of a learned 4x4 parameter; log-det 0
    (flow/rottrans.py:8-35).  The SVD (a 4x4, once per parameter version) is parameter preprocessing on the host, done with the
    same ``torch.svd`` call as the reference because U^T V depends on the SVD's arbitrary column signs; the per-sample arithmetic
    is the constant-matrix quaternion kernel."""

    _rnf_kind = runtime.KIND_AFFINE16
    _rnf_orthogonal = True

    def __init__(self):
        super().__init__()
        self.rot = nn.Parameter(torch.randn((1, 4, 4)) * 1e-3 + torch.eye(4).unsqueeze(0))
        self._cache = runtime.PackCache()

    def _rnf_pack(self, L, prec=0):
        with torch.no_grad():
            U, S, V = torch.svd(self.rot.detach().cpu().float())
            rot_mat = U.transpose(-1, -2) @ V
        return runtime.pack_rot16(L, rot_mat), None, 0, 0

    def _rnf_shape(self):
        return (self._rnf_kind, 0, 0)

    def _rnf_train_tensors(self):
        U, S, V = torch.svd(self.rot.cpu().float())      # differentiable; the 4x4 SVD stays on the host (see _rnf_pack)
        return [U.transpose(-1, -2) @ V]

    def forward(self, rotation, permute=None, feature=None):
        return self._single(rotation, permute, None, inverse=False)

    def inverse(self, rotation, permute=None, feature=None):
        return self._single(rotation, permute, None, inverse=True)

ConditionRot = _not_built("ConditionRot", "flow/rottrans.py:26-53")
Uncondition9RotL = _not_built("Uncondition9RotL", "flow/rottrans.py:94-105")
Condition9RotL = _not_built("Condition9RotL", "flow/rottrans.py:108-121")
Uncondition9RotR = _not_built("Uncondition9RotR", "flow/rottrans.py:124-135")
Condition9RotR = _not_built("Condition9RotR", "flow/rottrans.py:138-151")
Uncondition9RotRSmith = _not_built("Uncondition9RotRSmith", "flow/rottrans.py:154-165")
Condition9RotRSmith = _not_built("Condition9RotRSmith", "flow/rottrans.py:168-181")
