"""CPU: the C-ABI library loads and exports every symbol include/rnf_hip.h declares; host-side mirror of the reference
API (registry, layer order, state-dict keys, permutation schedule, error behaviour).  No compute calls (no GPU here)."""
import contextlib
import ctypes
import io
import os
import re

import numpy as np
import pytest
import torch

from oracle import flow_oracle as orc
from rotationnormflow_amd import _lib, make_config, runtime
from rotationnormflow_amd.flow import affineflow, squeezetrans
from rotationnormflow_amd.flow.flow import Flow, get_flow
from rotationnormflow_amd.flow.mobiusflow import MobiusFlow, get_mobius

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def quiet_flow(cfg):
    with contextlib.redirect_stdout(io.StringIO()):
        return Flow(cfg)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rnf_hip.h")).read()
    declared = set(re.findall(r"\b(rnf_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), name
    assert _lib.lib().rnf_abi_version() == _lib.ABI_VERSION == 7


def test_packed_sizes_match_parameter_counts():
    L = _lib.lib()
    assert L.rnf_mobius_packed_floats(64) == 29376           # == parameter count of an unconditional Moebius layer
    assert L.rnf_affine16_packed_floats() == 244
    assert L.rnf_featproj_packed_floats(256) == 64 * 256 + 64
    assert L.rnf_featproj_packed_floats(40) == 2 * 3 * 512 + 64          # k-steps of 16, zero padded (f16x2 image)
    assert L.rnf_workspace_bytes(1 << 20, 0) == 4096 * 8
    assert L.rnf_workspace_bytes(1 << 20, 25) == 4096 * 8 + 25 * 8192 * 2048 * 4


@pytest.mark.parametrize("preset,n_layers,n_keys", [("C1", 16, 88), ("C2", 48, 264), ("C4", 48, 273), ("C5", 42, 420), ("C5u", 42, 420)])
def test_flow_structure_and_state_dict_keys(preset, n_layers, n_keys):
    cfg = make_config(preset)
    fl = quiet_flow(cfg)
    sd = fl.state_dict()
    assert len(fl.layers) == n_layers and len(sd) == n_keys
    assert {k: tuple(v.shape) for k, v in sd.items()} == orc.state_shapes(cfg)      # oracle shapes are pinned to the reference's
    kinds = orc.layer_kinds(cfg)
    for layer, kind in zip(fl.layers, kinds):
        assert {"mobius": 1, "uncond16": 2, "cond16": 3}[kind] == layer._rnf_kind


def test_prints_layer_count_like_reference(capsys):
    Flow(make_config("C1"))
    assert "total layers of flow:  16" in capsys.readouterr().out


def test_permutation_schedules_forward_and_inverse():
    for preset in ("C2", "C4", "C5"):
        cfg = make_config(preset)
        fl = quiet_flow(cfg)
        fwd, inv = fl._forward_rows(), fl._inverse_rows()
        # restate flow.py:58-70 / 77-90 independently
        count, want = 0, []
        for k in orc.layer_kinds(cfg):
            want.append(count % 6)
            if k == "mobius" or cfg.frequent_permute:
                count += 1
        assert fwd == want
        for layer, a, b in zip(fl.layers, fwd, inv):
            if isinstance(layer, MobiusFlow):
                assert a == b


def test_registry_table():
    mk = lambda **kw: make_config(**kw)  # noqa: E731
    assert isinstance(affineflow.get_affine(mk(rot="16Trans"), 0), squeezetrans.Uncondition16Trans)
    assert isinstance(affineflow.get_affine(mk(rot="16Trans", condition=1), 32), squeezetrans.Condition16Trans)
    assert isinstance(affineflow.get_affine(mk(rot="16UnTrans", condition=1), 32), squeezetrans.Uncondition16Trans)
    assert isinstance(affineflow.get_affine(mk(rot="16UnTrans", condition=1), 32, first_layer_condition=True), squeezetrans.Condition16Trans)
    assert affineflow.get_affine(mk(rot="16UnTrans"), 0) is None          # unconditional table has no 16UnTrans (affineflow.py:48-73)
    assert affineflow.get_affine(mk(rot="None"), 0) is None
    assert affineflow.get_affine(mk(rot="None", condition=1), 8) is None
    assert get_mobius(mk(dist="noflow"), 0) is None
    from rotationnormflow_amd.flow import rottrans
    assert isinstance(affineflow.get_affine(mk(rot="16Trans", lu=1), 0), squeezetrans.Uncondition16TransLU)
    assert isinstance(affineflow.get_affine(mk(rot="16UnTrans", lu=1, condition=1), 8), squeezetrans.Uncondition16TransLU)
    assert isinstance(affineflow.get_affine(mk(rot="16Rot"), 0), rottrans.UnconditionRot)
    assert isinstance(affineflow.get_affine(mk(rot="16UnRot", condition=1), 8), rottrans.UnconditionRot)
    for rot, cls in (("36Trans", squeezetrans.Uncondition36Trans), ("9TransLSVD", rottrans.Uncondition9RotL),
                     ("9TransRSVD", rottrans.Uncondition9RotR), ("9TransLSmith", squeezetrans.Uncondition9Trans),
                     ("9TransRSmith", rottrans.Uncondition9RotRSmith)):
        assert isinstance(affineflow.get_affine(mk(rot=rot), 0), cls)
    for rot, cls in (("9TransLSVD", rottrans.Condition9RotL), ("9TransRSVD", rottrans.Condition9RotR),
                     ("9TransLSmith", squeezetrans.Condition9Trans), ("9TransRSmith", rottrans.Condition9RotRSmith),
                     ("36Trans", squeezetrans.Condition36Trans)):
        assert isinstance(affineflow.get_affine(mk(rot=rot, condition=1), 8), cls)
    # the conditional LU / SVD-rotation rows (built in round 2 as per-sample "side" matrices)
    assert isinstance(affineflow.get_affine(mk(rot="9TransLSmith", lu=1, condition=1), 8), squeezetrans.Condition9TransLU)
    assert isinstance(affineflow.get_affine(mk(rot="16Rot", condition=1), 8), rottrans.ConditionRot)
    assert isinstance(affineflow.get_affine(mk(rot="9TransLSmith", lu=1), 0), squeezetrans.Uncondition9TransLU)
    assert isinstance(affineflow.get_affine(mk(rot="16Trans", lu=1, condition=1), 8), squeezetrans.Condition16TransLU)
    assert isinstance(affineflow.get_affine(mk(rot="16UnTrans", lu=1, condition=1), 8, first_layer_condition=True), squeezetrans.Condition16TransLU)
    assert isinstance(affineflow.get_affine(mk(rot="16UnRot", condition=1), 8, first_layer_condition=True), rottrans.ConditionRot)
    lu = squeezetrans.Condition16TransLU(8)
    assert all(k.startswith("net.") for k in lu.state_dict()) and len(lu.state_dict()) == 5 + 3 * 10
    assert {"net.w_p", "net.u_mask", "net.l_mask", "net.s_sign", "net.l_eye", "net.w_l_net.fc_first.weight", "net.w_u_net.fc_last.bias",
            "net.w_s_net.layers.3.weight"} <= set(lu.state_dict())            # flow/squeezetrans.py:110-119 names


def test_rot_none_with_last_affine_is_a_type_error():
    with pytest.raises(TypeError):
        quiet_flow(make_config(layers=2, condition=1, feature_dim=8, rot="None", last_affine=1))


def test_cpu_tensors_are_refused_not_silently_computed():
    fl = quiet_flow(make_config(layers=1))
    with torch.no_grad(), pytest.raises(RuntimeError, match="GPU only"):
        fl(torch.eye(3)[None].repeat(4, 1, 1))


def test_pack_cache_follows_parameter_versions():
    fl = quiet_flow(make_config(layers=1))
    p1 = fl._packed("cpu")
    assert fl._packed("cpu") is p1
    with torch.no_grad():
        fl.layers[1].mat.add_(0.1)                     # in-place update, as an optimizer step / load_state_dict does
    p2 = fl._packed("cpu")
    assert p2 is not p1
    assert not np.array_equal(p1.blob.numpy(), p2.blob.numpy())


def test_rootfinder_order_is_a_flow_property_stamped_into_the_descriptor(tmp_path):
    """Round 6: the inverse root finder's first-pass order (include/rnf_hip.h desc column 5 bits 16..17) -- Flow.set_rootfinder_order, the
    checkpoint sidecar's "rootfinder_first_order" -- reaches the descriptor of the Moebius layers and nothing else."""
    from rotationnormflow_amd import harness
    fl = quiet_flow(make_config(layers=2))
    p = fl._packed("cpu")
    mob = p.desc[:, 0] == runtime.KIND_MOBIUS
    assert mob.any() and (~mob).any() and not (p.desc[:, 5] >> 16).any()
    before = p.desc.copy()
    fl.set_rootfinder_order(4)
    assert fl._packed("cpu") is p and ((p.desc[mob, 5] >> 16) & 3 == 2).all() and not (p.desc[~mob, 5] >> 16).any()
    fl.set_rootfinder_order(3)
    assert ((fl._packed("cpu").desc[mob, 5] >> 16) & 3 == 1).all()
    fl.set_rootfinder_order(None)
    assert np.array_equal(fl._packed("cpu").desc, before)
    with pytest.raises(ValueError):
        fl.set_rootfinder_order(5)
    ck = tmp_path / "ckpt.pth"
    assert harness.read_sidecar(ck) == {} and harness.read_feature_scale(ck) is None
    harness.write_feature_scale(ck, 4.0)
    harness.write_rootfinder_order(ck, 4)
    assert harness.read_sidecar(ck) == {"feature_mean_square": 4.0, "rootfinder_first_order": 4} and harness.read_feature_scale(ck) == 4.0
    harness.write_feature_scale(ck, 16.0)                            # one key rewritten, the other kept
    assert harness.read_sidecar(ck)["rootfinder_first_order"] == 4
    with pytest.raises(ValueError):
        harness.write_rootfinder_order(ck, 5)


def test_install_as_reference_modules():
    import sys

    import rotationnormflow_amd
    saved = {k: v for k, v in sys.modules.items() if k == "flow" or k.startswith("flow.") or k.startswith("utils")}
    try:
        rotationnormflow_amd.install_as_reference_modules()
        from flow.flow import Flow as F2, get_flow as g2          # what agent.py:9 imports
        from utils.fisher import MatrixFisherN as M2               # agent.py:10
        assert F2 is Flow and g2 is get_flow and M2.__name__ == "MatrixFisherN"
    finally:
        for k in [k for k in sys.modules if k == "flow" or k.startswith("flow.") or k.startswith("utils")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_fisher_constants_match_oracle():
    from rotationnormflow_amd import synth
    from rotationnormflow_amd.utils.fisher import MatrixFisherN
    for kind in ("diag531", "tilted"):
        A = torch.from_numpy(synth.fisher_A(kind))
        d = MatrixFisherN(A)
        R = torch.from_numpy(synth.uniform_rotations(64, seed=2)).double()
        want = orc.fisher_log_prob(R, A, torch.float64)
        got = (R * A.double().reshape(1, 3, 3)).sum((-1, -2)) - d.log_const().double()
        assert (got - want).abs().max() < 2e-6


def test_argument_validation_happens_before_any_gpu_work():
    """Every entry point checks its arguments on the host first: these calls return an error code (and a message) without a GPU."""
    import numpy as np
    L = _lib.lib()
    err = lambda: L.rnf_last_error().decode()  # noqa: E731
    buf = np.zeros(128, np.float32)
    # host packers
    assert L.rnf_pack_gs(buf.ctypes.data, 4, buf.ctypes.data) != 0 and "3 or 6" in err()
    assert L.rnf_pack_gs(np.zeros(9, np.float32).ctypes.data, 3, buf.ctypes.data) != 0 and "singular" in err()
    assert L.rnf_gs_packed_floats(3) == 20 and L.rnf_gs_packed_floats(6) == 72 and L.rnf_gs_packed_floats(5) == -1
    assert L.rnf_cond36_packed_floats() == L.rnf_cond16_packed_floats() + 2080
    # training entry points
    tdesc = np.array([[1, 0, 0]], np.int32)
    args = (None, None, 64, 0, None, tdesc.ctypes.data, 1, 513, None, None, None, None, None, None, None)      # K <= 512 (LDS of a 16-rotation block)
    assert L.rnf_flow_backward(*args) != 0 and "segments" in err()
    args = (None, None, 64, 0, None, tdesc.ctypes.data, 401, 64, None, None, None, None, None, None, None)        # <= 400 layers, like the forward passes
    assert L.rnf_flow_backward(*args) != 0 and "n_layers" in err()
    args = (None, None, 0, 0, None, tdesc.ctypes.data, 1, 64, None, None, None, None, None, None, None)
    assert L.rnf_flow_backward(*args) == 0                                  # empty batch: nothing to do
    pdesc = np.array([[15, 0, 0, -1]], np.int32)
    assert L.rnf_pack_flow_device(buf.ctypes.data, pdesc.ctypes.data, 1, 64, 0, 1, buf.ctypes.data, buf.ctypes.data, None) != 0 and "kind" in err()
    assert L.rnf_pack_flow_device(buf.ctypes.data, pdesc.ctypes.data, 1, 0, 0, 1, buf.ctypes.data, buf.ctypes.data, None) != 0 and "must be positive" in err()
    assert L.rnf_plain_layer_floats(1, 64, 0) == 29376 and L.rnf_plain_layer_floats(2, 64, 0) == 16
    assert L.rnf_plain_layer_floats(3, 64, 40) == 64 * 40 + 64 + 3 * 4160 + 16 * 65
    # shared feature rows
    assert L.rnf_workspace_bytes_shared(1 << 20, 42, 512) == 4096 * 8 + 42 * 2048 * 64 * 4
    assert L.rnf_workspace_bytes_shared(1 << 20, 42, 0) == L.rnf_workspace_bytes(1 << 20, 42)
    # round-2 entry points: plain-blob sizes of the new trainable kinds, side-layer training, the conditioner backward, the Fisher gradient
    assert L.rnf_plain_layer_floats(5, 64, 0) == 36 and L.rnf_plain_layer_floats(11, 64, 24) == 0
    assert L.rnf_plain_layer_floats(6, 64, 24) == 64 * 24 + 64 + 3 * 4160 + 9 * 65
    assert L.rnf_plain_layer_floats(10, 64, 24) == 64 * 24 + 64 + 3 * 4160 + 36 * 65
    sdesc = np.array([[11 | (0 << 16), 0, 0]], np.int32)
    args = (0, buf.ctypes.data, None, None, 64, 0, buf.ctypes.data, sdesc.ctypes.data, 1, 16, None, None, None, buf.ctypes.data, None, buf.ctypes.data,
            None, buf.ctypes.data, None)
    assert L.rnf_flow_backward_side(*args) != 0 and "side" in err()             # a side layer without its side / side_grad buffers
    args = (2,) + args[1:]
    assert L.rnf_flow_backward_side(*args) != 0 and "dir" in err()
    assert L.rnf_flow_train_side(3, None, None, 0, 0, None, None, None, 1, 8, None, None, None, None, 0, None) != 0 and "dir" in err()
    assert L.rnf_cond_mlp_backward(buf.ctypes.data, 64, 24, buf.ctypes.data, 65, buf.ctypes.data, None, None, buf.ctypes.data, None) != 0 and "n_out" in err()
    assert L.rnf_cond_mlp_backward(None, 64, 24, buf.ctypes.data, 16, buf.ctypes.data, None, None, buf.ctypes.data, None) != 0 and "null" in err()
    assert L.rnf_fisher_scratch_bytes(7) == (2 + 70) * 8
    assert L.rnf_fisher_log_const_nt(buf.ctypes.data, 3, 2, buf.ctypes.data, 16, buf.ctypes.data, None) != 0 and "norm_type" in err()
    assert L.rnf_fisher_log_const_nt(buf.ctypes.data, 3, 0, None, 0, buf.ctypes.data, None) != 0 and "scratch" in err()
    assert L.rnf_fisher_log_prob_backward_param(buf.ctypes.data, buf.ctypes.data, 10, buf.ctypes.data, 3, 1, buf.ctypes.data, 4096, buf.ctypes.data,
                                                None) != 0 and "divisible" in err()
    assert L.rnf_fisher_log_prob_backward_param(buf.ctypes.data, buf.ctypes.data, 9, buf.ctypes.data, 3, 1, buf.ctypes.data, 8, buf.ctypes.data,
                                                None) != 0 and "scratch" in err()
    # small kernels
    assert L.rnf_min_geodesic(None, None, 5, 0, None, None) != 0
    assert L.rnf_fisher_log_const(None, 3, None, None) != 0


def test_pmc_summary_tool(tmp_path):
    """tools/pmc_summary.py: mean per dispatch over the largest grid only, gfx950 x2 correction on FETCH_SIZE, KiB units."""
    import json
    import subprocess
    import sys
    d = tmp_path / "pass1"
    d.mkdir()
    head = ('"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name",'
            '"Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value",'
            '"Start_Timestamp","End_Timestamp"\n')
    rows = []
    for disp, grid, fetch, write in ((1, 4096, 10.0, 1.0), (2, 262144, 100.0, 8.0), (3, 262144, 300.0, 8.0)):
        for xcd_part in (0.25, 0.75):                                  # rocprofv3 emits one row per counter instance
            rows.append(f'{disp},{disp},"Agent 2",1,1,1,{grid},7,"void rnf::flow_stack_kernel<0, 0, 16, true, 1, false>(rnf::FlowArgs)",1024,0,0,64,64,96,'
                        f'"FETCH_SIZE",{fetch * xcd_part},1,2\n')
        rows.append(f'{disp},{disp},"Agent 2",1,1,1,{grid},7,"void rnf::flow_stack_kernel<0, 0, 16, true, 1, false>(rnf::FlowArgs)",1024,0,0,64,64,96,'
                    f'"WRITE_SIZE",{write},1,2\n')
        rows.append(f'{disp},{disp},"Agent 2",1,1,1,{grid},9,"other_kernel",64,0,0,8,0,16,"FETCH_SIZE",999.0,1,2\n')
    (d / "run_counter_collection.csv").write_text(head + "".join(rows))
    out_csv, out_json = tmp_path / "s.csv", tmp_path / "s.json"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), str(out_csv), str(out_json), "flow_stack_kernel", str(d)],
                   check=True, capture_output=True)
    s = json.loads(out_json.read_text())
    assert s["counters"]["FETCH_SIZE"] == 200.0 and s["counters"]["WRITE_SIZE"] == 8.0        # the two big-grid dispatches only
    assert s["hbm_fetch_bytes_per_launch"] == 200.0 * 1024 * 2 and s["hbm_write_bytes_per_launch"] == 8.0 * 1024
    assert s["workgroup_size"] == 1024


def test_pack_cache_sees_a_replaced_parameter_object():
    """replacing a Parameter OBJECT (module.weight = nn.Parameter(...)) leaves the old object's id / _version / data_ptr
    untouched; the cache must notice through its identity check of the live ``_parameters`` slots."""
    import torch
    from rotationnormflow_amd import runtime
    m = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 2))
    cache, calls = runtime.PackCache(), []

    def build():
        calls.append(1)
        return len(calls)
    assert cache.get(m, "cpu", build) == 1 and cache.get(m, "cpu", build) == 1
    m[0].weight = torch.nn.Parameter(torch.zeros(4, 3))
    assert cache.get(m, "cpu", build) == 2 and cache.get(m, "cpu", build) == 2
    with torch.no_grad():
        m[1].bias.add_(1.0)                                  # in-place edit: version bump
    assert cache.get(m, "cpu", build) == 3
    cache.invalidate()
    assert cache.get(m, "cpu", build) == 4


def test_parameter_and_state_dict_order_is_the_references(golden_dir):
    """``optim.Adam(flow.parameters())`` numbers its state by ``parameters()`` order and every checkpoint carries that numbering
    (agent.py:23,143,193-196): the mirrored modules must register their members in the reference's order (ConditionalTransform assigns
    fc_first, fc_last and THEN the ModuleList, flow/condition.py:13-22).  tests/golden/param_order.json holds, for every structure of the
    fixture tables, what the reference's own Flow yields (make_param_order.py).  Per-tensor flows and flattened flows alike."""
    import contextlib
    import io
    import json
    from rotationnormflow_amd.flow.flow import Flow
    with open(os.path.join(golden_dir, "param_order.json")) as fh:
        table = json.load(fh)
    assert len(table) >= 40
    flat_seen = 0
    for name, e in table.items():
        with contextlib.redirect_stdout(io.StringIO()):
            fl = Flow(make_config(**e["cfg"]))
        assert [[k, list(v.shape)] for k, v in fl.named_parameters()] == e["parameters"], name
        assert [[k, list(v.shape)] for k, v in fl.state_dict().items()] == e["state_dict"], name
        if fl.flatten_parameters():
            flat_seen += 1
            assert [[k, list(shape)] for k, _, shape in fl._flat_layout()] == e["parameters"], name
            assert [[k, list(v.shape)] for k, v in fl.state_dict().items()] == e["state_dict"], name
    assert flat_seen >= 10


def test_bench_compact_line_stays_parseable_and_small():
    """The driver parses the LAST stdout line of bench.py and keeps ~8 KB of stdout tail (round 4's 40 KB single line was lost): the compact
    record built from a committed full record (profiles/r5/bench_default.json) must stay under bench.COMPACT_LIMIT, carry every key of the
    bench contract with `roofline` and `cpu_baseline`, and shrink -- not overflow -- when more workloads are added."""
    import copy
    import json
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "r5", "bench_default.json")) as fh:
        full = json.load(fh)
    line = bench.compact_record(full, "bench_full.json")
    assert len(line) < bench.COMPACT_LIMIT and "\n" not in line
    rec = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in rec, k
    assert rec["config"]["workload"].startswith("C2") and rec["vs_baseline"] is None and rec["mean_nll"] == full["mean_nll"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(rec["roofline"]) and abs(rec["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-4
    assert {"value", "unit", "cores", "kind", "sample"} <= set(rec["cpu_baseline"])
    assert sorted(rec["configs"]) == sorted(full["configs"]) and rec["configs"]["C4q"]["traffic_x"] > 1
    # twenty more workloads: the line thins its per-workload entries, then drops optional blocks, and never loses the headline
    fat = copy.deepcopy(full)
    for i in range(20):
        fat["configs"][f"X{i}"] = copy.deepcopy(full["configs"]["C4"])
    line = bench.compact_record(fat, "bench_full.json")
    rec = json.loads(line)
    assert len(line) < bench.COMPACT_LIMIT and rec["value"] == json.loads(bench.compact_record(full))["value"] and "roofline" in rec
