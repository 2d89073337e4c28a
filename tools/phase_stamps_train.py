#!/usr/bin/env python3
"""Diagnostic: where does the backward sweep (csrc/train_block16.h, csrc/train_kernels.h) spend its cycles?  Builds a SEPARATE library with
-DRNF_STAMPS (the shipped librnf_hip.so contains no stamps), runs one training forward + backward and prints the share of
wave-0 cycles per phase.

    python tools/phase_stamps_train.py [--preset C2] [--batch 1024] [--block 16|64]
"""
import argparse
import contextlib
import io
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PHASES = ["0 load state + fc_first forward", "1 hidden layers forward", "2 fc_last forward", "3 layer math (segments) fwd+bwd",
          "4 fc_last weight gradient + bias", "5 fc_last data gradient", "6 hidden layers backward", "7 fc_first backward",
          "8 affine16 layer", "9 block epilogue"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="C2")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--block", type=int, default=16, choices=[16, 64], help="rotations per workgroup: csrc/train_block16.h or csrc/train_kernels.h")
    args = ap.parse_args()
    os.environ["RNF_TRAIN_BLOCK"] = str(args.block)
    out = os.path.join(ROOT, "tools", "_build", "librnf_hip_stamps.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    csrc = os.path.join(ROOT, "rotationnormflow_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc))
    if not os.path.exists(out) or os.path.getmtime(out) < newest:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-Xarch_host", "-mf16c",
                        "-DRNF_STAMPS", "-shared", "-fPIC", "-o", out, os.path.join(csrc, "rnf_api.hip")], check=True)
    import torch
    from rotationnormflow_amd import _lib, make_config, synth
    _lib.LIB_PATH = out
    from rotationnormflow_amd.flow.flow import Flow

    dev = torch.device("cuda", 0)
    cfg = make_config(args.preset)
    with contextlib.redirect_stdout(io.StringIO()):
        fl = Flow(cfg)
    shapes = {k: tuple(v.shape) for k, v in fl.state_dict().items()}
    fl.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=0).items()})
    fl = fl.to(dev).train()
    R = torch.from_numpy(synth.uniform_rotations(args.batch, seed=1)).to(dev)
    feat = torch.from_numpy(synth.features(args.batch, fl.feature_dim, seed=2)).to(dev) if cfg.condition else None
    stamps = torch.zeros(10, dtype=torch.int64, device=dev)
    os.environ["RNF_TRAIN_STAMPS_PTR"] = str(stamps.data_ptr())
    for it in range(2):
        fl.zero_grad()
        stamps.zero_()
        _, ldj = fl(R, feat)
        (-ldj).mean().backward()
        torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(float)
    blocks = (args.batch + args.block - 1) // args.block
    n_mlp = sum(1 for layer in fl.layers if hasattr(layer, "conditioner") or hasattr(layer, "net"))
    n_aff = len(fl.layers) - n_mlp
    print(f"preset={args.preset} batch={args.batch}: wave-0 shader-clock cycles (s_memtime), {blocks} workgroups, "
          f"{n_mlp} conditioner layers + {n_aff} constant affine layers")
    print(f"{'phase':40s} {'share':>8s} {'cycles/workgroup':>18s} {'cycles/layer':>14s}")
    for i, (name, v) in enumerate(zip(PHASES, s)):
        per = v / blocks / (n_aff if i == 8 else (1 if i == 9 else n_mlp)) if (n_aff if i == 8 else n_mlp) else 0.0
        print(f"{name:40s} {100 * v / s.sum():6.1f} %   {v / blocks:16.0f} {per:14.0f}")


if __name__ == "__main__":
    main()
