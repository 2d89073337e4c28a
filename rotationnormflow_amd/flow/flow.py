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
    """flow/flow.py:9-10 -- what the reference's drivers call (agent.py:20: ``self.flow = get_flow(config)``).  The flow comes back with its
    parameters FLATTENED (``Flow.flatten_parameters``: one ``nn.Parameter`` behind the unchanged state-dict keys) when its layer kinds
    allow it, because that is what makes the unedited training loop fast: ``optim.Adam(self.flow.parameters(), lr)`` (agent.py:23) then
    steps one tensor instead of 264, and autograd sees one leaf.  ``RNF_FLAT_PARAMS=0`` in the environment keeps per-tensor Parameters;
    ``Flow(config)`` itself never flattens."""
    import os
    flow = Flow(config)
    if os.environ.get("RNF_FLAT_PARAMS", "1") != "0":
        flow.flatten_parameters()
    return flow


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

    # ---- flat parameters (round 4) ----------------------------------------------------------------------------------
    def flatten_parameters(self) -> bool:
        """Move every parameter of the flow into ONE ``nn.Parameter`` (``self._flat``, fp32, the layout of the kernels' plain training
        blob = the reference's own tensor order) and leave the per-layer tensors behind as persistent BUFFERS that are views into it.
        What stays the same: ``state_dict()`` / ``load_state_dict()`` keys and values (agent.py:132-151,171-198), ``.cuda()`` / ``.to()``,
        ``nn.DataParallel``, evaluation, every result bit for bit.  What changes: ``flow.parameters()`` yields one tensor -- the
        reference's ``optim.Adam(flow.parameters(), lr)`` (agent.py:23) steps ONE tensor (its per-tensor host bookkeeping over 264 tensors
        was 10 ms per iteration with the default Adam, the whole iteration is 0.8 ms) -- autograd sees one leaf instead of 264, the training
        kernels read the parameter storage in place (no ``cat``) and ``loss.backward()`` leaves one gradient blob in ``_flat.grad``
        (``named_parameter_gradients()`` gives per-key views).  Optimizer STATE has one entry in memory; ``optimizer.state_dict()`` /
        ``load_state_dict()`` of any optimizer built over the flat parameter speak the reference's per-tensor layout (264 entries:
        agent.py:143,193-196 work unedited, in both directions) through the hooks of rotationnormflow_amd/flatopt.py.
        Returns False (and changes nothing) when a layer's training tensors are not its raw parameters (LU / SVD parameterisations, side
        layers), when the flow is already flat, or when it has no parameters."""
        if self.is_flat:
            return False
        slots, tensors = [], []
        for layer in self.layers:
            fn = getattr(layer, "_rnf_train_tensors", None)
            if fn is None or getattr(layer, "_rnf_side_layer", False):
                return False
            owners = {id(prm): (mod, name) for mod in layer.modules() for name, prm in mod._parameters.items() if prm is not None}
            for t in fn():
                if id(t) not in owners or t.dtype is not torch.float32:
                    return False                                   # a computed tensor (LU product, SVD rotation): stays classic
                slots.append(owners[id(t)])
                tensors.append(t)
            if sum(1 for _ in layer.parameters()) != len({id(t) for t in fn()}):
                return False                                       # a parameter that is not part of the training blob
        if not tensors:
            return False
        if len({bool(t.requires_grad) for t in tensors}) != 1 or len({t.device for t in tensors}) != 1:
            return False                                           # partly frozen flow (or parameters on several devices): stays per-tensor
        with torch.no_grad():
            flat = nn.Parameter(torch.cat([t.detach().reshape(-1) for t in tensors]), requires_grad=bool(tensors[0].requires_grad))
        # the blob is laid out in the kernels' order; ``parameters()`` order (= the numbering of a per-tensor optimizer's state, which
        # checkpoints carry: agent.py:143,193-196) is kept beside it
        rank = {id(prm): i for i, prm in enumerate(self.parameters())}
        self._flat_slots, self._flat_rank, off = [], [], 0
        for (mod, name), t in zip(slots, tensors):
            self._flat_rank.append(rank[id(t)])
            self._flat_slots.append((mod, name, off, tuple(t.shape)))
            off += t.numel()
        for r in sorted(range(len(slots)), key=lambda i: self._flat_rank[i]):      # buffers re-registered in parameters() order: state_dict() order
            mod, name = slots[r]
            del mod._parameters[name]
            mod.register_buffer(name, None)
        self.register_parameter("_flat", flat)
        self._realias()
        self.invalidate()
        # checkpoints keep the reference's per-tensor optimizer state (agent.py:143,193-196): rotationnormflow_amd/flatopt.py
        from .. import flatopt
        flatopt.install()
        return True

    def _flat_layout(self):
        """[(state-dict key, offset, shape)] of the slices of ``_flat``, in the reference's ``parameters()`` order."""
        names = {id(mod): name for name, mod in self.named_modules()}
        rank = getattr(self, "_flat_rank", None) or list(range(len(self._flat_slots)))
        order = sorted(range(len(self._flat_slots)), key=lambda i: rank[i])
        return [((names[id(mod)] + "." if names[id(mod)] else "") + name, off, shape)
                for mod, name, off, shape in (self._flat_slots[i] for i in order)]

    @property
    def is_flat(self) -> bool:
        return self._parameters.get("_flat") is not None

    def _realias(self):
        """Point every per-layer buffer at its slice of ``_flat``.  The slices are cut from ``_flat.detach()``: same storage and the SAME
        version counter as the parameter (an in-place write through either side is seen by the pack cache and by autograd's saved-tensor
        check), but no autograd view relation to the leaf (views of a leaf made under no_grad may not be touched again once the leaf has
        been written in place)."""
        from .. import flatopt
        flat = self._parameters["_flat"]
        base = flat.detach()
        for mod, name, off, shape in self._flat_slots:
            n = 1
            for d in shape:
                n *= d
            mod._buffers[name] = base[off:off + n].view(shape)
        flatopt.tag(flat, self._flat_layout())                     # (a deepcopy / unpickled Parameter is a new object: tagged again here)

    def _ensure_alias(self):
        """The per-layer views must alias ``_flat``; copy.deepcopy / pickling a module clones every tensor on its own, after which the copy's
        views would silently go stale.  Checked (one pointer comparison) wherever the views or the parameter are about to be used."""
        if getattr(self, "_is_replica", False):                    # nn.DataParallel replica: its slot table still names the MASTER's modules;
            return                                                 # a replica's buffers are broadcast copies and need no aliasing
        if self.is_flat and self._flat_slots:
            mod, name, off, _ = self._flat_slots[-1]
            flat = self._parameters["_flat"]
            buf = mod._buffers.get(name)
            if buf is None or buf.device != flat.device or buf.data_ptr() != flat.data_ptr() + 4 * off or not hasattr(flat, "_rnf_flat_layout"):
                self._realias()
                self.invalidate()

    def _replicate_for_data_parallel(self):
        replica = super()._replicate_for_data_parallel()
        # replica.__dict__ is a shallow copy of the master's: the guard watch (a pinned word and an event of the MASTER's device) must not be
        # shared across the worker threads / devices of nn.DataParallel (agent.py:22)
        replica.__dict__.pop("_rnf_guard_watch", None)
        return replica

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in self._TRANSIENT:
                new.__dict__[k] = copy.deepcopy(v, memo)
        new._ensure_alias()                                        # the cloned views alias the cloned parameter again
        return new

    _TRANSIENT = ("_rnf_train_plan", "_rnf_guard_watch")           # per-process launch state (pinned words, events): rebuilt on demand

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k not in self._TRANSIENT}

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._ensure_alias()
        if self.is_flat:
            from .. import flatopt
            flatopt.install()

    def _apply(self, fn, recurse=True):
        flat = self.is_flat and not getattr(self, "_is_replica", False)
        if flat and fn(torch.empty(0, dtype=torch.float32, device=self._parameters["_flat"].device)).dtype is not torch.float32:
            # checked BEFORE anything is converted, so that the module is not left half-way
            raise TypeError("a flattened flow keeps its parameters in fp32 (the kernels' plain blob); convert inputs, not the module")
        out = super()._apply(fn, recurse)
        if flat:
            self._realias()                                        # .cuda() / .to() moved the buffers as separate tensors
            self.invalidate()
        return out

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self._ensure_alias()
        super()._save_to_state_dict(destination, prefix, keep_vars)
        destination.pop(prefix + "_flat", None)                    # the reference's keys only: the values live in the per-layer views

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        self._ensure_alias()
        had = set(missing_keys)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        if prefix + "_flat" in missing_keys and prefix + "_flat" not in had:
            missing_keys.remove(prefix + "_flat")

    def named_parameter_gradients(self):
        """{state-dict key: gradient view or None} -- per-tensor views of ``_flat.grad`` for a flattened flow, the parameters' own
        ``.grad`` otherwise."""
        if not self.is_flat:
            return {k: p.grad for k, p in self.named_parameters()}
        g = self._parameters["_flat"].grad
        names = {id(mod): name for name, mod in self.named_modules()}
        out = {}
        for mod, name, off, shape in self._flat_slots:
            n = 1
            for d in shape:
                n *= d
            key = (names[id(mod)] + "." if names[id(mod)] else "") + name
            out[key] = None if g is None else g[off:off + n].view(shape)
        return out

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
        Default (round 6, ``runtime.set_feature_calibration("weights")``): 1 -- nothing is measured, the packed images are a function of the
        weights alone.  Features of another scale (un-normalised backbone outputs) are what this call, ``calibrate_feature_scale``, a checkpoint
        sidecar (``harness.write_feature_scale``) or -- sharded -- ``dist.calibrate_feature_scale`` (one all-reduced value on every rank) are
        for.  ``None`` returns to the default."""
        self._feature_ms_fixed = None if mean_square is None else runtime.quantise_feature_ms(float(mean_square))
        self.invalidate()

    def set_rootfinder_order(self, order):
        """Order of the FIRST pass of the inverse pass's root finder (csrc/flow_kernels.h mobius_inv_finish, the stand-in for BinFind,
        flow/mobiusflow.py:189-224): 3 (Halley; the default, ``None``) or 4 (Householder's step with the third derivative: four more
        instructions per segment pair in that pass, and then two passes instead of three for most waves when the conditioners' outputs are
        sharply peaked -- trained conditional flows: the reference-trained SYMSOL checkpoint inverts 3.7 % faster, synthetic weights 1 - 4 %
        slower).  A property of the flow (checkpoint sidecars carry it: ``harness.write_rootfinder_order``), never of a launch; the grid cell
        the root finder returns is the same for either."""
        if order not in (None, 3, 4):
            raise ValueError(f"root-finder first-pass order must be None, 3 or 4, got {order!r}")
        self._rnf_rf_order = order

    def calibrate_feature_scale(self, feature):
        """Measure the mean square of ``feature``'s entries (one device reduction, one scalar read-back) and fix it as the calibration
        (``set_feature_scale``).  Returns the value (quantised to 1/16 binade: what the packers see)."""
        self.set_feature_scale(runtime.feature_mean_square(feature))
        return self._feature_ms_fixed

    def _device_packed(self, device, feature=None):
        """Kernel blob built on the device from the live parameters (rnf_pack_flow_device, one 18 us launch): used when the host cache
        cannot be trusted or would thrash -- training mode (optimizers may write through .data) and nn.DataParallel replicas (fresh
        parameter tensors on every forward, agent.py:22).  None when a layer has no device-packing support (ragged K, 3x3 / 6x6 kinds)."""
        from .. import autograd
        layers = list(self.layers)
        flat = self._parameters.get("_flat")
        try:
            plan = autograd._plan_for(self, layers, self._forward_rows(), torch.empty(0, device=device))
            tensors = [flat] if flat is not None else autograd.train_tensors(layers)
        except NotImplementedError:
            return None
        with torch.no_grad():
            f32 = torch.float32
            # nn.DataParallel (agent.py:22) hands every replica broadcast copies of the parameters on ITS device and calls it from a worker
            # thread with that device's inputs.  A replica that still saw the master's storage would make this device read another GPU's
            # memory through the kernels (silently slow over xGMI, or a fault): refuse loudly on the first such forward instead.
            for t in ([flat] if flat is not None else tensors):
                if t.is_cuda and t.device != device:
                    raise RuntimeError(f"rotationnormflow_amd: parameters live on {t.device} but this call runs on {device} "
                                       f"({'a DataParallel replica' if getattr(self, '_is_replica', False) else 'module'}): "
                                       "move the module with .to(device) / let nn.DataParallel replicate it")
            if flat is not None and flat.is_cuda:
                plain = flat.detach()                              # flattened flow: the parameter storage IS the plain blob
            else:
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
                        if plan.feature_ms is None and runtime.get_feature_calibration() != "first-batch":
                            plan.feature_ms = 1.0
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
        packed = self._packed_images(device, feature)
        runtime.stamp_rootfinder_order(packed, getattr(self, "_rnf_rf_order", None))
        return packed

    def _packed_images(self, device, feature=None):
        """``feature``: the batch at hand; its mean square calibrates the equalisation of the conditional layers when a parameter version
        is packed (runtime.feature_mean_square; evaluations do not look at it again)."""
        if (self.training or getattr(self, "_is_replica", False)) and torch.device(device).type == "cuda":
            packed = self._device_packed(torch.device(device), feature)
            if packed is not None:
                return packed

        self._ensure_alias()
        watch = self.__dict__.get("_rnf_guard_watch")
        if watch is not None and self.condition and getattr(self, "_feature_ms_fixed", None) is None and watch.poll():
            # the guard kept firing: measure the feature scale again on this batch -- and re-pack only if the packers would see another value
            # (same value = same images: the cause is not the calibration, a repack would cost a D2H copy per poll for nothing)
            packed = self._cache.peek(device)
            if packed is None or feature is None or runtime.feature_mean_square(feature) != getattr(packed, "feature_ms", None):
                self._cache.invalidate()
                self._cache.feature_ms = None
            else:
                watch.recalibrations = watch.MAX_RECALIBRATIONS

        def build():
            rows = self._forward_rows()
            inv = self._inverse_rows()
            # Moebius layers are the only ones that read the row; forward and inverse schedules agree on them
            for layer, a, b in zip(self.layers, rows, inv):
                if isinstance(layer, MobiusFlow) and a % 3 != b % 3:
                    raise RuntimeError("forward/inverse permutation schedules disagree (flow/flow.py:58-90)")
            fixed = getattr(self, "_feature_ms_fixed", None)
            ms = 1.0 if not self.condition else (fixed if fixed is not None else runtime.calibration_ms(feature))
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
