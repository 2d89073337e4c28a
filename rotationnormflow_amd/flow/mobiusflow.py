"""Mirror of the reference's flow/mobiusflow.py: ``get_mobius`` and ``MobiusFlow``."""
import torch.nn as nn

from .. import runtime
from .condition import ConditionalTransform


def get_mobius(config, feature_dim):
    """flow/mobiusflow.py:7-14."""
    if config.dist == "noflow":
        return None
    return MobiusFlow(3, config.segments, condition=config.condition, feature_dim=feature_dim)


def _perm_row_of(permute):
    """The reference passes one row of its 6x3 table (a length-3 index tensor/sequence); rows are cyclic shifts."""
    p = [int(v) for v in permute]
    if p not in ([0, 1, 2], [1, 2, 0], [2, 0, 1]):
        raise NotImplementedError(f"permutation {p} is not a row of the reference's table (flow/flow.py:13-15)")
    return p[0]


class _SingleLayer:
    """Run one layer module through the fused-stack entry points (used when a layer is called on its own)."""

    def _single(self, rotation, permute, feature, inverse):
        row = 0 if permute is None else _perm_row_of(permute)

        def packed():
            p = self._cache.get(self, rotation.device, lambda: runtime.pack_layers([self], [row], rotation.device))
            if p.desc[0, 1] != row:
                p = runtime.pack_layers([self], [row], rotation.device)
            return p
        return runtime.run_flow(self, packed, rotation, feature, inverse=inverse, train_layers=[self], train_rows=[row])


class MobiusFlow(nn.Module, _SingleLayer):
    """Moebius coupling layer on one column of R conditioned on another (flow/mobiusflow.py:27-183)."""

    _rnf_kind = runtime.KIND_MOBIUS

    def __init__(self, D, K, condition=0, feature_dim=None):
        super().__init__()
        if D != 3:
            raise NotImplementedError("MobiusFlow is built for D=3 (the only value the reference uses)")
        self.D, self.K = D, K
        self.condition = condition
        self.feature_dim = feature_dim
        ni = D + (feature_dim if condition else 0)
        self.conditioner = ConditionalTransform(ni, 4 * K)
        self._cache = runtime.PackCache()

    def _rnf_pack(self, L, prec=0):
        F = self.feature_dim if self.condition else 0
        rec, frec = runtime.pack_mobius(L, self.conditioner, self.K, F, prec)
        return rec, frec, F, self.K

    def _rnf_shape(self):
        return (self._rnf_kind, self.K, self.feature_dim if self.condition else 0)

    def _rnf_train_tensors(self):
        from ..autograd import mlp_train_tensors
        return mlp_train_tensors(self.conditioner)

    def forward(self, rotation, permute=None, feature=None):
        assert permute is not None, "The permuting function is needed in this module"
        if self.condition:
            assert feature is not None, "The input feature is needed in this module"
        return self._single(rotation, permute, feature if self.condition else None, inverse=False)

    def inverse(self, trotation, permute=None, feature=None):
        assert permute is not None, "The permuting function is needed in this module"
        if self.condition:
            assert feature is not None, "feature input is needed in this module"
        return self._single(trotation, permute, feature if self.condition else None, inverse=True)
