"""Mirror of the reference's flow/condition.py: the conditioner MLP as a *parameter container*.

State-dict names match the reference exactly (``fc_first``, ``layers.{1,3,5}``, ``fc_last``; flow/condition.py:14-22)
so published checkpoints load unchanged.  The arithmetic (flow/condition.py:24-30) never runs in PyTorch: the owning
layer packs these tensors into the kernel blob (runtime.pack_mobius / pack_cond16) and the MLP executes on the fp32
matrix cores inside the fused stack kernel.
"""
import torch
import torch.nn as nn

from .. import _lib, runtime


class ConditionalTransform(nn.Module):
    """ConditionalTransform(Ni, No, Nh=64): Linear(Ni,64) -> [ReLU, Linear(64,64)] x3 -> ReLU(residual) -> Linear(64,No)."""

    def __init__(self, Ni, No, Nh=64):
        super().__init__()
        if Nh != 64:
            raise NotImplementedError("the HIP kernels are built for the reference's hidden width Nh=64")
        self.Ni, self.No = Ni, No
        # Registration ORDER is part of the checkpoint contract: the reference assigns fc_first, relu_last, fc_last and only then the
        # ModuleList (flow/condition.py:13-22), so ``parameters()`` yields fc_first, fc_last, layers.1/3/5 -- the numbering of the Adam state
        # every checkpoint carries (agent.py:23,143,193-196; pinned by tests/golden/param_order.json).
        self.fc_first = nn.Linear(Ni, Nh)
        self.relu_last = nn.ReLU()
        self.fc_last = nn.Linear(Nh, No)
        # same module indices as the reference (ReLU at 0/2/4, Linear at 1/3/5) => same state-dict keys
        self.layers = nn.ModuleList([nn.ReLU(), nn.Linear(Nh, Nh), nn.ReLU(), nn.Linear(Nh, Nh), nn.ReLU(), nn.Linear(Nh, Nh)])

    def forward(self, x):
        """Standalone evaluation on the GPU; supported for the unconditional Moebius conditioner shape (Ni=3, No=4K)."""
        if self.Ni != 3 or self.No % 32:
            raise NotImplementedError("standalone ConditionalTransform.forward is only built for Ni=3, No=4K (K%8==0); "
                                      "inside a flow the MLP runs fused in the stack kernel")
        if not x.is_cuda:
            raise RuntimeError("rotationnormflow_amd runs on the GPU only (no CPU fallback)")
        K = self.No // 4
        L = _lib.lib()
        prec = runtime._PRECISIONS[runtime.device_precision()]      # (the standalone conditioner kernel has no bf16x3 instantiation)
        try:
            rec, _ = runtime.pack_mobius(L, self, K, 0, prec)
        except runtime.HalfRangeError:
            prec = _lib.PREC_FP32
            rec, _ = runtime.pack_mobius(L, self, K, 0, prec)
        y = x.reshape(-1, 3).float().contiguous()
        blob = torch.from_numpy(rec).to(y.device)
        out = torch.empty(y.shape[0], self.No, dtype=torch.float32, device=y.device)
        with torch.cuda.device(y.device):
            _lib.check(L.rnf_conditioner_forward(y.data_ptr(), y.shape[0], blob.data_ptr(), K, prec, out.data_ptr(),
                                                 torch.cuda.current_stream(y.device).cuda_stream))
        return out
