"""Time one training iteration (forward + backward + Adam) of the reference's unconditional recipe (settings/raw.yml: 24 layers,
64 segments, batch 1024; train_uncondition.py / agent.py:75-92) on one MI355X, and the pieces it is made of.

    python tools/bench_train.py [--batch 1024] [--steps 20] [--config C2] [--cpu-oracle]

--cpu-oracle also times the same iteration with torch autograd of the fp32 oracle on the host cores (the "port" baseline)."""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rotationnormflow_amd import configs, synth  # noqa: E402
from rotationnormflow_amd.flow.flow import Flow  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--cpu-oracle", action="store_true")
    ap.add_argument("--graph", action="store_true", help="capture the iteration as a HIP graph (harness.GraphedTrainStep) and replay it")
    ap.add_argument("--fused-adam", action="store_true", help="torch.optim.Adam(fused=True): one optimizer launch instead of foreach kernels")
    ap.add_argument("--classic", action="store_true", help="per-tensor nn.Parameters (Flow(config)) instead of the flattened flow get_flow(config) returns")
    a = ap.parse_args()
    cfg = configs.make_config(a.config)
    with contextlib.redirect_stdout(io.StringIO()):
        fl = Flow(cfg)
        if not a.classic:
            fl.flatten_parameters()                        # what flow.flow.get_flow returns to the reference's drivers (agent.py:20)
    shapes = {k: tuple(v.shape) for k, v in fl.state_dict().items()}
    w = synth.fill_state_dict(shapes, seed=0)
    fl.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    fl = fl.cuda().train()
    opt = torch.optim.Adam(fl.parameters(), lr=1e-4, fused=True) if a.fused_adam else torch.optim.Adam(fl.parameters(), lr=1e-4)
    R = torch.from_numpy(synth.uniform_rotations(a.batch, seed=1)).cuda()
    feat = None
    if cfg.condition:
        feat = torch.from_numpy(synth.features(a.batch, fl.feature_dim, seed=2)).cuda()

    if a.graph:
        from rotationnormflow_amd.harness import GraphedTrainStep
        opt = torch.optim.Adam(fl.parameters(), lr=1e-4, fused=True, capturable=True)
        gstep = GraphedTrainStep(fl, opt, tuple(R.shape), None if feat is None else tuple(feat.shape))
        for _ in range(a.warmup):
            gstep(R, feat)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = gstep(R, feat)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print(json.dumps(dict(metric="training iteration (forward + backward + Adam), HIP graph replay", optimizer="Adam(fused, capturable)",
                              config=a.config, batch=a.batch, ms_per_iteration=dt * 1e3, rotations_per_s=a.batch / dt,
                              loss=float(loss.detach()))))
        return

    def step():
        opt.zero_grad(set_to_none=True)
        _, ldj = fl(R, feat)
        loss = (-ldj).mean()
        loss.backward()
        opt.step()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    # pieces
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ev[0].record()
    _, ldj = fl(R, feat)
    ev[1].record()
    loss = (-ldj).mean()
    ev[2].record()
    loss.backward()
    ev[3].record()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    with torch.no_grad():
        next(fl.parameters()).add_(0.0)                      # new parameter version -> the next call repacks
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    fl._packed(R.device)
    torch.cuda.synchronize()
    pack_ms = (time.perf_counter() - t3) * 1e3
    out = dict(parameters="flat (1 tensor)" if fl.is_flat else f"classic ({len(list(fl.parameters()))} tensors)", pack_ms=pack_ms, optimizer="Adam(fused=True)" if a.fused_adam else "Adam", metric="training iteration (forward + backward + Adam)", config=a.config, batch=a.batch, ms_per_iteration=dt * 1e3,
               rotations_per_s=a.batch / dt, forward_ms_gpu=ev[0].elapsed_time(ev[1]), backward_ms_gpu=ev[2].elapsed_time(ev[3]),
               fwd_bwd_wall_ms=(t2 - t1) * 1e3, loss=float(loss.detach()))
    if a.cpu_oracle:
        from oracle import flow_oracle as orc
        p = {k: torch.from_numpy(v).requires_grad_(True) for k, v in w.items()}
        opt_o = torch.optim.Adam(list(p.values()), lr=1e-4)
        Rc = R.cpu()
        fc = None if feat is None else feat.cpu()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            opt_o.zero_grad()
            _, l = orc.flow_forward(cfg, p, Rc, fc, dtype=torch.float32, grad=True)
            (-l).mean().backward()
            opt_o.step()
            ts.append(time.perf_counter() - t0)
        out["cpu_oracle_ms_per_iteration"] = min(ts) * 1e3
        out["cpu_threads"] = torch.get_num_threads()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
