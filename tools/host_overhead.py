#!/usr/bin/env python3
"""Host-side cost of one call through the Python mirror + ctypes C ABI at the training batch of the recipe (N = 1024, settings/raw.yml):
wall time per call with the GPU kept busy-free (synchronised) vs. the GPU time of the same call (HIP events), and the pieces of the
host path (cache key, input checks, ctypes launch).  `python tools/host_overhead.py` (GPU box).
`--train`: host time of the five statements of one eager training iteration (agent.py:83-90: zero_grad, forward, loss, backward, step) for
the flattened flow `get_flow` returns and for per-tensor parameters, plain and fused Adam."""
import contextlib
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rotationnormflow_amd import make_config, runtime, synth  # noqa: E402
from rotationnormflow_amd.flow.flow import Flow  # noqa: E402


def train_profile():
    from rotationnormflow_amd.flow.flow import get_flow
    dev = torch.device("cuda", 0)
    cfg = make_config("C2")
    R = torch.from_numpy(synth.uniform_rotations(1024, seed=1)).to(dev)
    names = ["zero_grad", "forward", "loss", "backward", "step"]
    for label, make in (("flattened (get_flow)", get_flow), ("per tensor (Flow)", Flow)):
        for fused in (False, True):
            with contextlib.redirect_stdout(io.StringIO()):
                fl = make(cfg)
            fl = fl.to(dev).train()
            opt = torch.optim.Adam(fl.parameters(), lr=1e-4, fused=fused)
            warm, reps = 10, 40
            acc = [0.0] * 5
            for it in range(warm + reps):
                if it == warm:
                    torch.cuda.synchronize()
                    t_start, acc = time.perf_counter(), [0.0] * 5
                t = [time.perf_counter()]
                opt.zero_grad(set_to_none=True); t.append(time.perf_counter())
                _, ldj = fl(R); t.append(time.perf_counter())
                loss = (-ldj).mean(); t.append(time.perf_counter())
                loss.backward(); t.append(time.perf_counter())
                opt.step(); t.append(time.perf_counter())
                for i in range(5):
                    acc[i] += t[i + 1] - t[i]
            torch.cuda.synchronize()
            total = (time.perf_counter() - t_start) / reps
            print(json.dumps({"parameters": label, "tensors": sum(1 for _ in fl.parameters()), "optimizer": "Adam(fused=True)" if fused else "Adam",
                              "ms_per_iteration": round(total * 1e3, 3), "host_ms": dict(zip(names, [round(a / reps * 1e3, 3) for a in acc]))}), flush=True)


def main():
    if "--train" in sys.argv:
        return train_profile()
    dev = torch.device("cuda", 0)
    cfg = make_config("C2")
    with contextlib.redirect_stdout(io.StringIO()):
        fl = Flow(cfg)
    w = synth.fill_state_dict({k: tuple(v.shape) for k, v in fl.state_dict().items()}, seed=1, regime="trained")
    fl.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    fl = fl.to(dev).eval()
    for n in (1024, 1 << 16):
        R = torch.from_numpy(synth.uniform_rotations(n, seed=2)).to(dev)
        with torch.no_grad():
            for _ in range(20):
                fl(R)
            torch.cuda.synchronize()
            reps = 200
            t0 = time.perf_counter()
            for _ in range(reps):
                fl(R)
            issue = (time.perf_counter() - t0) / reps          # host time to ISSUE a call (launches queue up)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fl(R)
                torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / reps           # call + wait
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fl(R)
            b.record()
            torch.cuda.synchronize()
            gpu = a.elapsed_time(b) / reps * 1e-3
            t0 = time.perf_counter()
            for _ in range(reps):
                fl._packed(dev)
            key = (time.perf_counter() - t0) / reps
        print(json.dumps({"n": n, "host_issue_us": issue * 1e6, "call_and_wait_us": wall * 1e6, "gpu_us_per_call_back_to_back": gpu * 1e6,
                          "pack_cache_lookup_us": key * 1e6, "parameters": sum(1 for _ in fl.parameters())}))


if __name__ == "__main__":
    main()
