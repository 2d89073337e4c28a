"""ctypes binding of librnf_hip.so (include/rnf_hip.h).  There is no fallback: if the HIP library has not been built
the import of any op fails loudly (build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``python -m rotationnormflow_amd.build``)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librnf_hip.so")

_lib = None
ABI_VERSION = 7
PREC_FP32, PREC_F16X2, PREC_BF16X3 = 0, 1, 2

c_f32p = C.c_void_p      # device or host float*, passed as integer addresses
c_i32p = C.c_void_p

_SIGNATURES = {
    "rnf_abi_version": (C.c_int, []),
    "rnf_last_error": (C.c_char_p, []),
    "rnf_last_pack_audit": (C.c_double, []),
    "rnf_set_equalize": (C.c_int, [C.c_int]),
    "rnf_set_feature_ms": (C.c_double, [C.c_double]),
    "rnf_set_pack_audit": (C.c_int, [C.c_int]),
    "rnf_set_fused": (C.c_int, [C.c_int]),
    "rnf_set_train_block": (C.c_int, [C.c_int]),
    "rnf_mobius_packed_floats": (C.c_int64, [C.c_int32]),
    "rnf_mobius_packed_floats_prec": (C.c_int64, [C.c_int32, C.c_int32]),
    "rnf_cond_packed_floats_prec": (C.c_int64, [C.c_int32, C.c_int32]),
    "rnf_affine16_packed_floats": (C.c_int64, []),
    "rnf_cond16_packed_floats": (C.c_int64, []),
    "rnf_featproj_packed_floats": (C.c_int64, [C.c_int32]),
    "rnf_pack_mobius": (C.c_int, [c_f32p] * 10 + [C.c_int32, C.c_int32, C.c_int32, c_f32p, c_f32p]),
    "rnf_pack_affine16": (C.c_int, [c_f32p, c_f32p]),
    "rnf_pack_rot16": (C.c_int, [c_f32p, c_f32p]),
    "rnf_gs_packed_floats": (C.c_int64, [C.c_int32]),
    "rnf_pack_gs": (C.c_int, [c_f32p, C.c_int32, c_f32p]),
    "rnf_pack_cond16": (C.c_int, [c_f32p] * 10 + [C.c_int32, C.c_int32, c_f32p, c_f32p]),
    "rnf_pack_cond9": (C.c_int, [c_f32p] * 10 + [C.c_int32, C.c_int32, c_f32p, c_f32p]),
    "rnf_cond36_packed_floats": (C.c_int64, []),
    "rnf_pack_cond36": (C.c_int, [c_f32p] * 10 + [C.c_int32, C.c_int32, c_f32p, c_f32p]),
    "rnf_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "rnf_workspace_bytes_segments": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "rnf_flow_forward": (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_int32, c_f32p, c_i32p, C.c_int32, C.c_int32,
                                   c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnf_flow_inverse": (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_int32, c_f32p, c_i32p, C.c_int32, C.c_int32,
                                   c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnf_flow_forward_side": (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_int32, c_f32p, c_f32p, c_i32p, C.c_int32, C.c_int32,
                                        c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnf_flow_inverse_side": (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_int32, c_f32p, c_f32p, c_i32p, C.c_int32, C.c_int32,
                                        c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnf_flow_log_prob_side": (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_int32, c_f32p, c_f32p, c_i32p, C.c_int32, C.c_int32,
                                         c_f32p, c_f32p, C.c_int64, c_f32p, c_f32p, c_f32p, C.c_void_p,
                                         C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnf_cond_mlp_forward": (C.c_int, [c_f32p, C.c_int64, C.c_int32, c_f32p, C.c_int32, C.c_int32, C.c_int32, c_f32p,
                                       C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnf_condrot_matrices": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_void_p, C.c_void_p]),
    "rnf_condrot_svd": (C.c_int, [c_f32p, C.c_int64, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_void_p]),
    "rnf_condlu_matrices": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, c_f32p, C.c_int32, c_f32p,
                                      C.c_void_p]),
    "rnf_condlu_backward": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p,
                                      c_f32p, c_f32p, C.c_void_p]),
    "rnf_pack_flow_device": (C.c_int, [c_f32p, c_i32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_f32p, c_i32p, C.c_void_p]),
    "rnf_plain_layer_floats": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "rnf_flow_forward_train": (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_int32, c_f32p, c_i32p, C.c_int32, C.c_int32,
                                         c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnf_flow_forward_train_plain": (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_int32, c_f32p, c_i32p, C.c_int32, C.c_int32,
                                               c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "rnf_train_acts_floats": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "rnf_flow_backward_saved": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32, c_f32p, c_i32p, C.c_int32, C.c_int32,
                                          c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "rnf_flow_backward": (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_int32, c_f32p, c_i32p, C.c_int32, C.c_int32,
                                    c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "rnf_flow_inverse_train": (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_int32, c_f32p, c_i32p, C.c_int32, C.c_int32,
                                         c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnf_flow_inverse_backward": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32, c_f32p, c_i32p, C.c_int32, C.c_int32,
                                            c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "rnf_workspace_bytes_shared": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int64]),
    "rnf_flow_forward_shared": (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_int64, c_f32p, c_i32p, C.c_int32, C.c_int32,
                                          c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnf_flow_inverse_shared": (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_int64, c_f32p, c_i32p, C.c_int32, C.c_int32,
                                          c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnf_flow_log_prob_shared": (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_int64, c_f32p, c_i32p, C.c_int32, C.c_int32,
                                           c_f32p, c_f32p, C.c_int64, c_f32p, c_f32p, c_f32p, C.c_void_p,
                                           C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnf_flow_log_prob": (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_int32, c_f32p, c_i32p, C.c_int32, C.c_int32,
                                    c_f32p, c_f32p, C.c_int64, c_f32p, c_f32p, c_f32p, C.c_void_p,
                                    C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnf_fisher_log_prob": (C.c_int, [c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int64, c_f32p, C.c_void_p]),
    "rnf_min_geodesic": (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_int32, c_f32p, C.c_void_p]),
    "rnf_fisher_log_const": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_void_p]),
    "rnf_fisher_proper_svd": (C.c_int, [c_f32p, C.c_int64, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "rnf_fisher_log_prob_backward": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p, C.c_void_p]),
    "rnf_flow_train_side": (C.c_int, [C.c_int32, c_f32p, c_f32p, C.c_int64, C.c_int32, c_f32p, c_f32p, c_i32p, C.c_int32, C.c_int32, c_f32p, c_f32p,
                                      c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnf_flow_backward_side": (C.c_int, [C.c_int32, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32, c_f32p, c_i32p, C.c_int32, C.c_int32, c_f32p, c_f32p,
                                         c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "rnf_cond_mlp_backward": (C.c_int, [c_f32p, C.c_int64, C.c_int32, c_f32p, C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "rnf_matrix_to_quaternion": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_void_p]),
    "rnf_fisher_scratch_bytes": (C.c_size_t, [C.c_int64]),
    "rnf_fisher_log_const_mc": (C.c_int, [c_f32p, C.c_int64, C.c_int64, C.c_uint64, C.c_void_p, C.c_size_t, c_f32p, C.c_void_p]),
    "rnf_fisher_log_const_nt": (C.c_int, [c_f32p, C.c_int64, C.c_int32, C.c_void_p, C.c_size_t, c_f32p, C.c_void_p]),
    "rnf_fisher_log_prob_backward_param": (C.c_int, [c_f32p, c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int32, C.c_void_p, C.c_size_t, c_f32p,
                                                  C.c_void_p]),
    "rnf_fisher_sample": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int64, C.c_uint64, c_f32p, C.c_void_p, C.c_void_p]),
    "rnf_conditioner_forward": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int32, C.c_int32, c_f32p, C.c_void_p]),
}

EXPORTS = tuple(_SIGNATURES)


def load(path):
    """Load one build of the library and bind every export (tools/ab_variants.py loads several builds side by side)."""
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the MI355X HIP library has not been built and there is no CPU fallback. "
            "Run `python -m rotationnormflow_amd.build` (needs hipcc).")
    # torch first: it maps its own HIP runtime (libamdhip64.so.7 under torch/lib), which the dynamic loader then also binds this library to.
    # Loaded the other way round (e.g. build() and smoke() in ONE process) the library would pull the system's runtime from /opt/rocm and the
    # process would hold two HIP runtimes, of which only torch's has the device open ("no ROCm-capable device is detected" in ours).
    import torch  # noqa: F401
    handle = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(handle, name)
        fn.restype = res
        fn.argtypes = args
    if handle.rnf_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path}: ABI version mismatch; rebuild")
    return handle


def lib():
    """The loaded library (loads on first use)."""
    global _lib
    if _lib is None:
        _lib = load(LIB_PATH)
    return _lib


def check(rc: int):
    if rc != 0:
        raise RuntimeError("librnf_hip: " + lib().rnf_last_error().decode())
