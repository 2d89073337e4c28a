"""Mirror of the reference's flow/flow.py: ``get_flow`` and ``Flow``.

``Flow(config)`` builds the same ModuleList (same order, same state-dict keys) as flow/flow.py:19-51; ``forward`` /
``inverse`` keep the reference signatures and return ``(rotation' [N,3,3], ldjs [N])`` but the whole layer stack runs
as ONE fused HIP kernel launch (rnf_flow_forward / rnf_flow_inverse).  ``log_prob`` is the fused density evaluation
(flow + matrix-Fisher base + NLL sum) used by the benchmark and by rotationnormflow_amd.dist.
"""
import torch
from torch import nn

from .. import runtime
from .affineflow import get_affine
from .mobiusflow import MobiusFlow, get_mobius


def get_flow(config):
    return Flow(config)


_permute_prop = torch.tensor(runtime.PERMUTE_ROWS, dtype=torch.long)     # flow/flow.py:13-15


class Flow(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.condition = config.condition
        self._permute = _permute_prop
        n_blocks = config.layers

        if self.condition:                                               # flow/flow.py:29-34
            self.feature_dim = 32 if config.feature_dim is None else config.feature_dim
            if config.embedding:
                self.feature_dim += config.embedding_dim
        else:
            self.feature_dim = 0

        stack = []
        if config.last_affine:                                           # flow/flow.py:37-39 (appended unguarded)
            stack.append(get_affine(config, self.feature_dim, first_layer_condition=True))
        for i in range(n_blocks):                                        # flow/flow.py:41-48
            mob = get_mobius(config, self.feature_dim)
            if mob is not None:
                stack.append(mob)
            aff = get_affine(config, self.feature_dim)
            if aff is not None and (i != n_blocks - 1 or config.first_affine):
                stack.append(aff)
        print("total layers of flow: ", len(stack))                      # flow/flow.py:50
        if any(layer is None for layer in stack):
            # the reference appends None here and dies with TypeError at the first call (flow.py:38,65)
            raise TypeError(f"rot={config.rot!r} with last_affine=1 yields no first affine layer ('NoneType' object is not callable)")
        self.layers = nn.ModuleList(stack)
        self._cache = runtime.PackCache()

    # ---- permutation schedule (flow/flow.py:58-70 and 77-90) ------------------------------------------------------
    def _forward_rows(self):
        rows, count = [], 0
        for layer in self.layers:
            rows.append(count % 6)
            if isinstance(layer, MobiusFlow) or self.config.frequent_permute:
                count += 1
        return rows

    def _inverse_rows(self):
        rows = [0] * len(self.layers)
        count = len(self.layers) if self.config.frequent_permute else self.config.layers
        for i in reversed(range(len(self.layers))):
            if isinstance(self.layers[i], MobiusFlow) or self.config.frequent_permute:
                count -= 1
            rows[i] = count % 6
        return rows

    def invalidate(self):
        """Drop every packed copy of the parameters.  Call it after editing parameters in a way ``Tensor._version`` does not record --
        ``p.data.copy_()`` / ``p.data.add_()``, EMA weight swaps, a fused / foreach optimizer stepped outside this module's backward --
        while the module is in eval mode.  (In training mode, and for nn.DataParallel replicas, every call re-packs on the device from
        the live parameters, so nothing can go stale there.)"""
        self._cache.invalidate()
        self._cache.feature_ms = None
        self.__dict__.pop("_rnf_train_plan", None)

    def set_feature_scale(self, mean_square):
        """Fix the calibration input of the conditional layers' pack-time equalisation (csrc/equalize.h): the mean square of a feature entry.
        By default it is measured on the first feature batch a parameter version is packed for -- per PROCESS, so the ranks of a sharded
        evaluation would each measure their own shard; ``dist.calibrate_feature_scale`` all-reduces one value and sets it here on every rank,
        which makes the packed images, and with them every rotation's result, identical on 1 and on N GPUs.  ``None`` returns to measuring."""
        self._feature_ms_fixed = None if mean_square is None else runtime.quantise_feature_ms(float(mean_square))
        self.invalidate()

    def _device_packed(self, device, feature=None):
        """Kernel blob built on the device from the live parameters (rnf_pack_flow_device, one 18 us launch): used when the host cache
        cannot be trusted or would thrash -- training mode (optimizers may write through .data) and nn.DataParallel replicas (fresh
        parameter tensors on every forward, agent.py:22).  None when a layer has no device-packing support (ragged K, 3x3 / 6x6 kinds)."""
        from .. import autograd
        layers = list(self.layers)
        try:
            plan = autograd._plan_for(self, layers, self._forward_rows(), torch.empty(0, device=device))
            tensors = autograd.train_tensors(layers)
        except NotImplementedError:
            return None
        with torch.no_grad():
            f32 = torch.float32
            plain = torch.cat([t.reshape(-1) if (t.is_cuda and t.dtype is f32) else t.to(device=device, dtype=f32).reshape(-1)
                               for t in tensors]) if tensors else torch.zeros(0, device=device)
            with torch.cuda.device(device):
                if self.condition:
                    # one measurement per flow, not per replica / per call: nn.DataParallel replicas are rebuilt every forward, but they
                    # share this cache object (runtime.PackCache)
                    fixed = getattr(self, "_feature_ms_fixed", None)
                    if fixed is not None:
                        plan.feature_ms = fixed
                    elif plan.feature_ms is None:
                        plan.feature_ms = getattr(self._cache, "feature_ms", None)
                    plan.calibrate(feature)
                    self._cache.feature_ms = plan.feature_ms
                blob = plan.pack(plain, torch.cuda.current_stream(device).cuda_stream, with_fallback=runtime._guard_fallback)
        desc = autograd.desc_with_fallback(plan) if runtime._guard_fallback else plan.desc
        packed = runtime.PackedFlow(blob, desc, plan.n_cond, plan.feat_dim, plan.feat_padded, plan.segments, plan.precision)
        # layers whose per-sample matrices the host builds (ConditionRot / ConditionLU): in side-slot order, as runtime.pack_layers does
        packed.side_layers = [layer for layer in layers if layer._rnf_kind in runtime.SIDE_KINDS]
        if packed.side_layers and not packed.feat_dim:
            packed.feat_dim = packed.side_layers[0].feature_dim
            packed.feat_padded = runtime.pad8(packed.feat_dim)
        return packed

    def _packed(self, device, feature=None):
        """``feature``: the batch at hand; its mean square calibrates the equalisation of the conditional layers when a parameter version
        is packed (runtime.feature_mean_square; evaluations do not look at it again)."""
        if (self.training or getattr(self, "_is_replica", False)) and torch.device(device).type == "cuda":
            packed = self._device_packed(torch.device(device), feature)
            if packed is not None:
                return packed

        def build():
            rows = self._forward_rows()
            inv = self._inverse_rows()
            # Moebius layers are the only ones that read the row; forward and inverse schedules agree on them
            for layer, a, b in zip(self.layers, rows, inv):
                if isinstance(layer, MobiusFlow) and a % 3 != b % 3:
                    raise RuntimeError("forward/inverse permutation schedules disagree (flow/flow.py:58-90)")
            fixed = getattr(self, "_feature_ms_fixed", None)
            ms = 1.0 if not self.condition else (fixed if fixed is not None else runtime.feature_mean_square(feature))
            return runtime.pack_layers(list(self.layers), rows, device, feature_ms=ms)
        return self._cache.get(self, device, build)

    # ---- reference API ----------------------------------------------------------------------------------------------
    def forward(self, rotation, feature=None, inverse=False, draw=False, feature_repeat=None):
        """``feature_repeat`` = Q (extension): ``feature`` holds N / Q rows, row r conditioning rotations [r Q, (r + 1) Q) -- what the
        reference expresses by materialising ``feature.repeat`` (agent.py:240-244); evaluation only."""
        if inverse:
            return self.inverse(rotation, feature, draw, feature_repeat=feature_repeat)
        if not self.condition:
            feature = None
        return runtime.run_flow(self, lambda: self._packed(rotation.device, feature), rotation, feature, inverse=False,
                                train_layers=list(self.layers), train_rows=self._forward_rows(), feature_repeat=feature_repeat)

    def inverse(self, rotation, feature=None, draw=False, feature_repeat=None):
        if not self.condition:
            feature = None
        return runtime.run_flow(self, lambda: self._packed(rotation.device, feature), rotation, feature, inverse=True,
                                train_layers=list(self.layers), train_rows=self._inverse_rows(), feature_repeat=feature_repeat)

    # ---- fused density evaluation (agent.py:54-65,217-229 + utils/fisher.py:217-232) -------------------------
    def log_prob(self, rotation, feature=None, base=None, return_rotation=False, feature_repeat=None):
        """Per-sample log p(R) = ldj + base(R') and {sum, count} in fp64, in one launch.
        ``base``: None (uniform) or a ``MatrixFisherN``.  Returns dict(logp, sum, rotation, ldj)."""
        if not self.condition:
            feature = None
        if runtime._needs_grad(rotation, feature, self) or (base is not None and torch.is_grad_enabled() and base.A.requires_grad):
            # a gradient is required (agent.py:54-65 inside train_func): the same quantity from the differentiable pieces -- the training
            # forward (states saved for the one-launch backward) and MatrixFisherN._log_prob (differentiable w.r.t. R' and A)
            rot_out, ldj = self.forward(rotation, feature, feature_repeat=feature_repeat)      # (shared rows are expanded under autograd)
            logp = ldj if base is None else ldj + base._log_prob(rot_out)
            total = torch.stack([logp.double().sum(), torch.tensor(float(logp.numel()), dtype=torch.float64, device=logp.device)])
            return dict(logp=logp, sum=total, rotation=rot_out if return_rotation else None, ldj=None)
        A = c = None
        if base is not None:
            A, c = base.A, base.log_const()
        return runtime.run_log_prob(self, self._packed(rotation.device, feature), rotation, feature, A, c,
                                    want_rotation=return_rotation, want_ldj=False, want_logp=True, feature_repeat=feature_repeat)
