"""CPU-only tests: the C ABI loads and exports every declared symbol, argument validation (no launches), the drop-in
module trees mirror the reference's state_dict layout, the seeded generator is frozen, frame sharding + collectives
(gloo, world_size 2)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, default_opts, install_dropin
from e4s2024_amd import seeded


# ------------------------------------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    from e4s2024_amd import _lib
    L = _lib.lib()
    declared = _lib.declared_symbols()
    assert len(declared) >= 22 and len(set(declared)) == len(declared)
    for name in declared:
        assert hasattr(L.cdll, name), f"{name} declared in include/e4s_hip.h but not exported by {L.path}"
    assert set(_lib._PROTOS) | {"e4s_abi_version", "e4s_last_error"} == set(declared)
    assert L.cdll.e4s_abi_version() == 1
    out = subprocess.run(["nm", "-D", "--defined-only", L.path], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert {s for s in exported if s.startswith("e4s_")} == set(declared)          # nothing undeclared leaks out either


def test_abi_signatures_carry_no_torch_types():
    import re
    src = open(os.path.join(ROOT, "include", "e4s_hip.h")).read()
    assert 'extern "C"' in src
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)            # declarations only, comments stripped
    assert "torch" not in code.lower() and "at::" not in code and "Tensor" not in code and "#include <hip" not in code
    for decl in re.findall(r"E4S_API\s+(?:int|const char\*)\s+e4s_\w+\([^;]+;", code):
        for arg in decl[decl.index("(") + 1: decl.rindex(")")].split(","):
            ty = " ".join(arg.split()[:-1])
            assert ty in ("", "void", "int", "float", "int64_t", "void*", "float*", "const float*", "uint8_t*", "const uint8_t*", "int*", "const int*", "int32_t*", "const int32_t*",
                          "const float* const*", "uint16_t*", "const uint16_t*", "int64_t*", "const void*", "const E4sStyleJob*", "const E4sChainLayer*", "unsigned"), (decl.split("(")[0], arg)


def test_argument_validation_without_gpu():
    """Rejected arguments return E4S_ERR_ARG with a message before anything is launched."""
    from e4s2024_amd._lib import lib
    L = lib()
    c = L.cdll
    one = ctypes.c_void_p(16)          # non-null dummy pointers: validation fails before they are touched
    assert c.e4s_upfirdn2d(one, one, one, 1, 4, 4, 40, 4, 1, 1, 1, 1, 0, 0, 0, 0, None) == -1
    assert b"not in 1..32" in c.e4s_last_error()
    assert c.e4s_upfirdn2d(one, one, one, 1, 4, 4, 4, 4, 0, 1, 1, 1, 0, 0, 0, 0, None) == -1
    assert c.e4s_upfirdn2d(one, one, one, 1, 2, 2, 4, 4, 1, 1, 1, 1, 0, 0, 0, 0, None) == -1          # empty output
    assert c.e4s_region_modconv3x3(one, one, one, one, None, None, 0, 0, None, 0, None, None, 0, 1, 8, 8, 4, 4, 12, 0, None, 0, None) == -1
    assert b"label map" in c.e4s_last_error()
    assert c.e4s_region_modconv3x3(one, one, one, one, None, one, 4, 4, None, 0, None, None, 0, 1, 8, 8, 4, 4, 17, 0, None, 0, None) == -1
    assert c.e4s_conv2d(one, one, None, 0, one, None, None, None, None, None, 0, 1, 8, 8, 4, 4, 5, 1, 2, None) == -1
    assert b"not supported" in c.e4s_last_error()
    assert c.e4s_conv2d(one, one, None, 0, one, None, None, None, None, None, 0, 1, 8, 8, 4, 4, 3, 3, 1, None) == -1
    assert c.e4s_modconv_prep_weights(one, None, one, None, 8, 8, 3, 1, None) == -1                   # up-conv without blur
    assert c.e4s_onehot_to_labels(one, None, one, 1, 17, 4, 4, None) == -1
    assert c.e4s_bicubic_down_normalize(one, one, one, None, None, 1, 3, 64, 64, 3, None) == -1
    assert c.e4s_fused_bias_act(None, None, None, None, 3, 0, 0.2, 1.0, 0, 1, 0, None) == 0           # empty tensor: nothing to do
    with pytest.raises(RuntimeError, match="e4s_upfirdn2d failed"):
        L.call("e4s_upfirdn2d", one, one, one, 1, 4, 4, 40, 4, 1, 1, 1, 1, 0, 0, 0, 0, None)


def test_ops_refuse_cpu_tensors_and_wrong_dtype():
    from e4s2024_amd import ops
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ops.fused_leaky_relu(torch.zeros(2, 3), torch.zeros(3))
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ops.upfirdn2d(torch.zeros(1, 1, 4, 4), torch.ones(2, 2))
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ops.mask_to_labels(torch.zeros(1, 12, 8, 8))
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ops.plane_stats(torch.zeros(1, 4, 8, 8), 1e-5)


# ------------------------------------------------------------------------------------------------ drop-in trees
def _meta(fn):
    with torch.device("meta"):
        return fn()


def test_dropin_state_dicts_match_reference_manifests(manifest):
    install_dropin()
    from models.networks import Net3, Net, LocalMLP
    from models.stylegan2.model import Generator
    from swap_face_fine.face_parsing.model import BiSeNet
    assert Net is Net3
    for name, build in (("net3_1024_rli13", lambda: Net3(default_opts())),
                        ("generator_64_rli5", lambda: Generator(64, 512, 8, split_layer_idx=5, remaining_layer_idx=5)),
                        ("generator_256_rli13", lambda: Generator(256, 512, 8, split_layer_idx=5, remaining_layer_idx=13)),
                        ("bisenet_19", lambda: BiSeNet(19))):
        sd = _meta(build).state_dict()
        ref = manifest[name]
        keys = [k if not name.startswith("generator") else "G." + k for k in sd]
        assert keys == list(ref.keys()), name                      # same keys in the same order
        for k, kk in zip(sd, keys):
            assert list(sd[k].shape) == ref[kk][0] and str(sd[k].dtype).replace("torch.", "") == ref[kk][1], (name, k)
    net = _meta(lambda: Net3(default_opts()))
    assert sorted(k for k, p in net.named_parameters() if not p.requires_grad) == manifest["net3_1024_rli13_requires_grad_false"]
    net = _meta(lambda: Net3(default_opts(train_G=True)))
    assert sorted(k for k, p in net.named_parameters() if not p.requires_grad) == manifest["net3_1024_rli13_trainG_requires_grad_false"]


def test_dropin_api_surface():
    """Names and signatures the unchanged callers touch."""
    import inspect
    install_dropin()
    from models.networks import Net3
    from models.stylegan2.model import Generator, StyledConv, ToRGB, ModulatedConv2d, EqualLinear
    import models.stylegan2.op as op
    from swap_face_fine.face_parsing.face_parsing_demo import init_faceParsing_pretrained_model, faceParsing_demo, vis_parsing_maps, FaceParser
    from swap_face_fine.face_parsing.model import BiSeNet, seg_mean, seg_std
    assert list(inspect.signature(Net3.gen_img).parameters) == ["self", "struc_codes", "style_codes", "mask", "randomize_noise", "noise", "return_latents"]
    assert list(inspect.signature(Net3.get_style_vectors).parameters) == ["self", "img", "mask"]
    assert list(inspect.signature(Net3.cal_style_codes).parameters) == ["self", "style_vectors"]
    assert list(inspect.signature(Net3.forward).parameters) == ["self", "img", "mask", "resize", "randomize_noise", "return_latents"]
    assert list(inspect.signature(Generator.forward).parameters) == ["self", "styles", "structure_feats", "mask", "return_latents", "inject_index",
                                                                      "truncation", "truncation_latent", "input_is_latent", "noise",
                                                                      "randomize_noise", "use_structure_code"]
    assert list(inspect.signature(StyledConv.forward).parameters)[:5] == ["self", "input", "style", "mask", "noise"]   # + engine-internal kwarg
    assert list(inspect.signature(ToRGB.forward).parameters) == ["self", "input", "style", "mask", "skip"]
    assert list(inspect.signature(ModulatedConv2d.forward).parameters) == ["self", "input", "style"]
    assert list(inspect.signature(op.fused_leaky_relu).parameters) == ["input", "bias", "negative_slope", "scale"]
    assert list(inspect.signature(op.upfirdn2d).parameters) == ["input", "kernel", "up", "down", "pad"]
    assert list(inspect.signature(faceParsing_demo).parameters) == ["model", "img", "convert_to_seg12"]
    assert seg_mean.device.type == "cpu" and tuple(seg_std.shape) == (1, 3, 1, 1)
    g = _meta(lambda: Generator(1024, 512, 8, split_layer_idx=5, remaining_layer_idx=13))
    assert [c.mask_op for c in g.convs] == [True] * 12 + [False] * 4
    assert [r.mask_op for r in g.to_rgbs] == [True] * 5 + [False] * 3 and g.conv1.mask_op and g.to_rgb1.mask_op
    assert g.n_latent == 18 and g.num_layers == 17


def test_install_only_overrides_hot_path_modules():
    import e4s2024_amd
    install_dropin()
    import importlib
    assert importlib.util.find_spec("models.encoders.model_irse") is None       # not ours: resolves from the reference tree when present
    assert set(e4s2024_amd.OVERRIDES) == {"models.networks", "models.stylegan2.model", "models.stylegan2.op", "models.stylegan2.op.fused_act",
                                          "models.stylegan2.op.upfirdn2d", "models.stylegan2.op.conv2d_gradfix", "models.encoders.psp_encoders",
                                          "swap_face_fine.face_parsing.model", "swap_face_fine.face_parsing.resnet",
                                          "swap_face_fine.face_parsing.face_parsing_demo"}
    import models.networks
    assert models.networks.__file__.startswith(e4s2024_amd.DROPIN_DIR)


def test_conv2d_gradfix_resolves_under_the_redirected_op_package():
    """SURVEY §1 lists conv2d_gradfix.{conv2d, conv_transpose2d} in the L1 interface (reference op/conv2d_gradfix.py:22-75; caller criteria/adv_loss.py:4)."""
    install_dropin()
    from models.stylegan2.op import conv2d_gradfix
    import torch.nn.functional as F
    x, w = torch.randn(2, 6, 5, 5), torch.randn(4, 3, 3, 3)
    assert torch.equal(conv2d_gradfix.conv2d(x, w, padding=1, groups=2), F.conv2d(x, w, padding=1, groups=2))
    wt = torch.randn(6, 2, 3, 3)
    assert torch.equal(conv2d_gradfix.conv_transpose2d(x, wt, stride=2, padding=0, groups=2), F.conv_transpose2d(x, wt, stride=2, groups=2))
    with conv2d_gradfix.no_weight_gradients():
        assert conv2d_gradfix.weight_gradients_disabled
    assert not conv2d_gradfix.weight_gradients_disabled


def test_remap_lut_matches_oracle():
    install_dropin()
    from swap_face_fine.face_parsing.face_parsing_demo import remap_lut
    from oracle import e4s_oracle as O
    assert (remap_lut()[:19] == O.remap_19_to_12(np.arange(19, dtype=np.uint8))).all() and remap_lut()[19:].max() == 0


# ------------------------------------------------------------------------------------------------ seeded generator
def test_seeded_stream_is_frozen():
    a = seeded.seeded_array(4, "G.conv1.conv.weight", (3, 5))
    assert abs(float(a.sum()) - (-7.7163143)) < 1e-5 and a.dtype == np.float32, float(a.sum())
    b = seeded.seeded_array(1, "codes", (2, 3), dist="normal")
    assert abs(float(b[0, 0]) - 1.9986119) < 1e-6, float(b[0, 0])
    lab = seeded.blocky_labels(3, 1, 12, 512, 16)
    assert lab.shape == (1, 512, 512) and int(lab.astype(np.int64).sum()) == 1432576, int(lab.astype(np.int64).sum())
    oh = seeded.labels_to_onehot(lab, 12)
    assert oh.sum().item() == 512 * 512 and (oh.argmax(1).numpy() == lab).all()


def test_bench_flop_accounting():
    sys.path.insert(0, ROOT)
    import bench
    fl = bench.conv3x3_flops_per_face()
    rgb = 2 * 3 * sum(c * r * r for c, r in ((512, 4), (512, 8), (512, 16), (512, 32), (512, 64), (256, 128), (128, 256), (64, 512), (32, 1024)))
    assert abs((sum(fl.values()) + rgb) / 1e9 - 148.52) < 0.05


def test_bench_compact_line_fits_the_drivers_window_whatever_the_detail_holds():
    """bench.py prints ONE line on stdout and the driver keeps its last 2 000 characters: the compact line must stay below that with every secondary leg present,
    with legs that failed (error strings), and with none; it carries the contract fields and both headline metrics of BASELINE.json as flat scalars."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    long = "x" * 900
    base = {"metric": "1024x1024 faces/sec (StyleGAN2 regional synthesis, gen_img)", "value": 12345.678, "unit": "faces/s", "n_gpus": 8, "steps": 20, "warmup": 5,
            "ms_per_step": 12.345, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16+mxfp6x2 (masked 3x3 layers >= 32^2) / bf16x3 (all other layers)",
            "data": "synthetic", "config": {"workload": long, "batch_per_gpu": 4, "global_batch": 32, "parallelism": "frames x8", "streams_per_gpu": 2, "step_overlap": long}}
    roof = {"bound": "mfma", "kernel": "region_modconv_mx_kernel<1>", "kernel_launches": long, "achieved": 1234.56, "peak": 1500.0, "unit": "TFLOP/s", "frac": 0.1234,
            "traffic": 273230028, "traffic_source": long, "peak_basis": long, "whole_job_frac": 0.1909, "launches_per_step": 7, "avg_launch_ms": 0.1834,
            "algorithmic_gflop_per_launch": 44.177, "in_overlapped_region": {"avg_launch_ms": 0.3032}, "by_layer": [{"layer": long}] * 7, "in_run_ab": {"what": long},
            "all_modconv3x3": {"by_kernel_ms_per_step": {"region_modconv_mx_kernel<1>": 1.3, "modconv_up_hc": 0.55, "chain_conv3x3<32>": 0.29}}}
    full = dict(base, roofline=roof, one_stream={"faces_per_s": 1294.9, "ms_per_step": 3.089, "what": long}, soak={"faces_per_s": 1407.4, "board": {"power_w": 1223.0, "sclk_mhz": 2046.0}},
                cpu_baseline={"value": 0.3661, "unit": "faces/s", "cores": 16, "host_hardware_threads": 256, "kind": "port", "sample": long, "max_abs_pixel_diff_vs_gpu": 1.997e-4},
                full_swap={"p50_ms_per_frame": 2.115, "swaps_per_s": 472.8, "overlapped_batches": {"ms_per_frame": 2.008}, "roofline": {"frac": 0.2402, "peak_basis": long},
                           "parity": {"max_abs_pixel_diff_vs_oracle": 2.066e-4, "parser_label_flips_vs_oracle": 0, "how": long}, "unit_of_work": long},
                pti={"s_per_iter": 0.01213, "roofline": {"frac": 0.0441, "what": long}, "how": long, "clip_loop": {"what": long}},
                clip={"frames_per_s": 469.9, "ms_per_frame": 2.128, "unit_of_work": long, "collectives_in_timed_region": long},
                mask_sensitivity={"portrait_like_ellipses": {"faces_per_s": 1278.9}, "coarse_4x4_cells": {"faces_per_s": 1367.7}, "iid_per_pixel": {"faces_per_s": 1276.9}},
                f16_range={"overflowed_in_the_measured_passes": False, "what": long})
    ksum = {k: (20, v * 20) for k, v in roof["all_modconv3x3"]["by_kernel_ms_per_step"].items()}
    failed = dict(full, full_swap=None, pti={"error": long}, clip={"error": long}, mask_sensitivity=None, cpu_baseline=None, soak=None)
    bare = dict(base, roofline=None, one_stream=None, soak=None, cpu_baseline=None, full_swap=None, pti=None, clip=None, mask_sensitivity=None, f16_range={})
    for line, one, ks in ((full, full["one_stream"], ksum), (failed, full["one_stream"], ksum), (bare, None, None)):
        out = bench._compact_line(line, one, ks, 20, "gpurun_out/bench_detail.json")
        text = json.dumps(out)
        assert len(text) < 2000, len(text)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                  "roofline", "cpu_baseline", "full_swap_p50_ms_per_frame", "pti_s_per_iter", "clip_frames_per_s"):
            assert k in out, k
        assert "workload" in out["config"] and "model" not in out["config"]
    out = bench._compact_line(full, full["one_stream"], ksum, 20, None)
    assert out["full_swap_p50_ms_per_frame"] == 2.115 and out["pti_s_per_iter"] == 0.01213 and out["stage_ms"]["ge512"] == round(0.55 + 0.29, 3)
    assert out["roofline"]["frac_at_soak_sclk"] == round(0.1234 * 2400.0 / 2046.0, 4) and out["soak_joules_per_face"] == round(1223.0 / 1407.4, 3)


# ------------------------------------------------------------------------------------------------ frame sharding (gloo, world 2)
def test_shard_range_partitions_every_frame_once():
    from e4s2024_amd.runner import shard_range
    for n in (0, 1, 2, 5, 8, 255, 256, 257):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                s, e = shard_range(n, r, world)
                assert 0 <= s <= e <= n
                seen += list(range(s, e))
                for i in range(s, e):
                    assert i * world // n == r
            assert seen == list(range(n))


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[3])
from e4s2024_amd.runner import FrameShardRunner, shard_range
rank, world, n_frames = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[4])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[5], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
r = FrameShardRunner()
shared = r.broadcast_shared(torch.arange(24, dtype=torch.float32).reshape(1, 2, 3, 4) if rank == 0 else None, (1, 2, 3, 4))
calls = []
def frame_inputs(lo, hi):
    return torch.arange(lo, hi)
def synth(shared, idx):
    calls.append((int(idx[0]) if len(idx) else -1, len(idx)))
    f = (idx.view(-1, 1, 1, 1) * 7 + shared.sum().long()) % 251
    return f.to(torch.uint8).expand(-1, 4, 5, 3).contiguous()
out = r.run_clip(n_frames, shared, frame_inputs, synth, batch=2)
t = r.max_over_ranks(float(rank + 1))
assert t == float(world)
# the streamed variant (one asynchronous gather per batch round, what bench.py --clip times), batch sizes that do and do not divide the blocks
streamed = [r.run_clip_streamed(n_frames, shared, frame_inputs, synth, batch=b) for b in (2, 3, 16)]
pre = torch.full((n_frames, 4, 5, 3), 255, dtype=torch.uint8) if rank == 0 else None
streamed.append(r.run_clip_streamed(n_frames, shared, frame_inputs, synth, batch=2, out=pre))
if rank == 0:
    exp = ((torch.arange(n_frames).view(-1, 1, 1, 1) * 7 + 276) % 251).to(torch.uint8).expand(-1, 4, 5, 3)
    assert out is not None and tuple(out.shape) == (n_frames, 4, 5, 3) and torch.equal(out, exp), (out.shape,)
    for st in streamed:
        assert st is not None and tuple(st.shape) == (n_frames, 4, 5, 3) and torch.equal(st, exp)
    assert streamed[-1] is pre
    print("RANK0_OK", n_frames)
else:
    assert out is None and all(st is None for st in streamed)
s, e = shard_range(n_frames, rank, world)
assert sum(c[1] for c in calls if c[0] >= s) >= e - s
dist.destroy_process_group()
'''


@pytest.mark.parametrize("n_frames,port", [(7, 29611), (1, 29612), (8, 29613)])
def test_run_clip_world2_gloo(tmp_path, n_frames, port):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", ROOT, str(n_frames), str(port)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert f"RANK0_OK {n_frames}" in outs[0]


_SYNC_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[3])
from e4s2024_amd import pti
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[4], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
unused = torch.nn.Parameter(torch.ones(4))                       # no gradient on any rank: skipped like on one GPU (no zero gradient)
only0 = torch.nn.Parameter(torch.ones(2))                        # gradient on rank 0 only
params = list(net.parameters()) + [unused, only0]
xs, ys = torch.randn(world, 4, 6), torch.randn(world, 4, 3)
opt = torch.optim.SGD(params, lr=0.1)
loss = torch.nn.functional.mse_loss(net(xs[rank]), ys[rank]) + (only0.sum() * 3 if rank == 0 else 0)
opt.zero_grad(); loss.backward()
n = pti.sync_gradients(params, bucket_bytes=64)                  # tiny buckets: several all-reduces in flight
assert n >= 3, n
ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
ref.load_state_dict(net.state_dict())
sum(torch.nn.functional.mse_loss(ref(xs[r]), ys[r]) for r in range(world)).div(world).backward()
for a, b in zip(net.parameters(), ref.parameters()):
    assert torch.allclose(a.grad, b.grad, atol=1e-6), (a.grad - b.grad).abs().max()
assert unused.grad is None
assert torch.allclose(only0.grad, torch.full((2,), 3.0 / world))
assert pti.sync_gradients(params, bucket_bytes=1 << 30) == 1
print("SYNC_OK", rank)
dist.destroy_process_group()
'''


def test_pti_gradient_average_world2_gloo(tmp_path):
    """the one exchange step of multi-GPU PTI — bucketed flat all-reduce of the gradients — gives every rank the gradient of
    the rank-averaged loss, also for a parameter that got no gradient on some ranks; one that got none on ANY rank is left alone (as a
    single-GPU step leaves it: no zero gradient, hence no optimiser state for it)."""
    script = tmp_path / "sync_worker.py"
    script.write_text(_SYNC_WORKER)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", ROOT, "29617"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "SYNC_OK 0" in outs[0] and "SYNC_OK 1" in outs[1]
    from e4s2024_amd import pti
    assert pti.sync_gradients([torch.nn.Parameter(torch.ones(2))]) == 0      # no process group: nothing to do


_TUNE_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[3])
from e4s2024_amd import pti
from e4s2024_amd.runner import shard_range
rank, world, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[5])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[4], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3)); ref.load_state_dict(net.state_dict())
xs, ys = torch.randn(n, 1, 6), torch.randn(n, 1, 3)
opt, ropt = torch.optim.SGD(net.parameters(), lr=0.1), torch.optim.SGD(ref.parameters(), lr=0.1)
calls = []
def step_fn(net_, opt_, vec, mp, img, fg, group, active):            # vec = x, img = y of this rank's frame (None: no frame left this round)
    opt_.zero_grad(set_to_none=True)
    loss = None
    if vec is not None:
        loss = torch.nn.functional.mse_loss(net_(vec[0]), img[0]); loss.backward()
    calls.append((vec is not None, active))
    pti.sync_gradients(list(net_.parameters()), group, active_ranks=active)
    opt_.step()
    return loss
hist = pti.tune_clip(net, opt, ys, torch.zeros(n, 1, 1, dtype=torch.uint8), xs, steps=2, group=None, step_fn=step_fn)
# the same schedule in one process: round i of a pass = the i-th frame of every rank's block, gradients averaged over the ranks that have one
blocks = [shard_range(n, r, world) for r in range(world)]
rounds = max(e - s for s, e in blocks)
for _ in range(2):
    for i in range(rounds):
        fr = [s + i for s, e in blocks if s + i < e]
        ropt.zero_grad()
        (sum(torch.nn.functional.mse_loss(ref(xs[f]), ys[f]) for f in fr) / len(fr)).backward()
        ropt.step()
for a, b in zip(net.parameters(), ref.parameters()):
    assert torch.allclose(a, b, atol=1e-6), (a - b).abs().max()
assert len(calls) == 2 * rounds and len(hist) == 2
lo, hi = blocks[rank]
assert [c[0] for c in calls[:rounds]] == [lo + i < hi for i in range(rounds)]
print("TUNE_OK", rank, calls[:rounds])
dist.destroy_process_group()
'''


@pytest.mark.parametrize("n_frames,port", [(5, 29621), (4, 29622), (1, 29623)])
def test_tune_clip_rounds_world2_gloo(tmp_path, n_frames, port):
    """BASELINE configs[3] on several GPUs: `pti.tune_clip` shards the clip's frames in blocks, walks them in rounds (one frame per rank per
    round, gradients averaged over the ranks that still have one) and keeps every rank's parameters identical to the single-process
    evaluation of the same schedule — also when the blocks are uneven (a rank sits a round out but still joins the collectives)."""
    script = tmp_path / "tune_worker.py"
    script.write_text(_TUNE_WORKER)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", ROOT, str(port), str(n_frames)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "TUNE_OK 0" in outs[0] and "TUNE_OK 1" in outs[1]


def test_run_clip_single_process_equals_sharded_result():
    from e4s2024_amd.runner import FrameShardRunner
    r = FrameShardRunner()
    out = r.run_clip(5, torch.ones(1), lambda lo, hi: torch.arange(lo, hi), lambda sh, idx: idx.view(-1, 1, 1, 1).to(torch.uint8).expand(-1, 2, 2, 3).contiguous(), batch=4)
    assert out[:, 0, 0, 0].tolist() == [0, 1, 2, 3, 4]
    out = r.run_clip_streamed(5, torch.ones(1), lambda lo, hi: torch.arange(lo, hi), lambda sh, idx: idx.view(-1, 1, 1, 1).to(torch.uint8).expand(-1, 2, 2, 3).contiguous(), batch=2)
    assert out[:, 0, 0, 0].tolist() == [0, 1, 2, 3, 4]


def test_handoff_files_roundtrip(tmp_path):
    """Row f4: a clip dumped with the reference's file names / encodings reads back exactly (masks, style vectors) or to the uint8
    quantisation of ``tensor2im`` (images)."""
    import numpy as np
    import torch
    from e4s2024_amd import handoff
    rs = np.random.RandomState(0)
    n = 3
    clip = handoff.ClipBatch(
        driven=torch.from_numpy(rs.uniform(-1, 1, (n, 3, 32, 32)).astype(np.float32)),
        target=torch.from_numpy(rs.uniform(-1, 1, (n, 3, 32, 32)).astype(np.float32)),
        driven_mask=torch.from_numpy(rs.randint(0, 12, (n, 16, 16)).astype(np.uint8)),
        target_mask=torch.from_numpy(rs.randint(0, 12, (n, 16, 16)).astype(np.uint8)),
        driven_style=torch.from_numpy(rs.standard_normal((n, 12, 1280)).astype(np.float32)),
        target_style=torch.from_numpy(rs.standard_normal((n, 12, 1280)).astype(np.float32)))
    handoff.dump(clip, str(tmp_path), first_index=5)
    assert sorted(os.listdir(tmp_path / "mask"))[0] == "D_mask_0005.png" and (tmp_path / "styleVec" / "T_style_vec_0007.pt").exists()
    back = handoff.load(str(tmp_path), first_index=5, size=32)
    assert len(back) == n
    assert torch.equal(back.driven_mask, clip.driven_mask) and torch.equal(back.target_mask, clip.target_mask)
    assert torch.equal(back.driven_style, clip.driven_style) and torch.equal(back.target_style, clip.target_style)
    q = (((clip.target.clamp(-1, 1) + 1) / 2 * 255).to(torch.uint8).float() / 255 - 0.5) / 0.5
    assert torch.allclose(back.target, q, atol=1e-6)
    assert torch.load(tmp_path / "styleVec" / "D_style_vec_0005.pt").shape == (1, 12, 1280)


def test_multi_band_blend_oracle_properties():
    """The oracle's restatement of cv2.pyrDown / pyrUp / the Laplacian blend (unpinned: cv2 is not in this image) at least has the
    properties the algorithm guarantees: constants survive both pyramid steps, the 8-bit step rounds like (sum + 128) >> 8, mask = 1
    returns A exactly, mask = 0 returns B, and sizes follow (n + 1) // 2 and 2 n."""
    from oracle import e4s_oracle as O
    rs = np.random.RandomState(3)
    c = np.full((9, 6, 2), 7.0)
    assert np.allclose(O.pyr_down(c), 7.0) and O.pyr_down(c).shape == (5, 3, 2)
    assert np.allclose(O.pyr_up(c), 7.0) and O.pyr_up(c).shape == (18, 12, 2)
    assert (O.pyr_down(np.full((8, 8, 1), 200, np.uint8)) == 200).all()
    a = rs.randint(0, 256, (8, 8, 1)).astype(np.uint8)
    d = O.pyr_down(a)
    k = np.array([1, 4, 6, 4, 1])
    pad = np.pad(a[..., 0].astype(np.int64), 2, mode="reflect")
    exp = np.array([[(np.outer(k, k) * pad[2 * y:2 * y + 5, 2 * x:2 * x + 5]).sum() + 128 >> 8 for x in range(4)] for y in range(4)])
    assert np.array_equal(d[..., 0], exp)
    A = rs.randint(0, 256, (64, 64, 3)).astype(np.uint8)
    B = rs.rand(64, 64, 3) * 255
    assert np.array_equal(np.clip(O.laplacian_blend(A, B, np.ones((64, 64, 3), np.float32), 6), 0, 255).astype(np.uint8), A)
    assert np.abs(O.laplacian_blend(A, B, np.zeros((64, 64, 3), np.float32), 6) - B).max() <= 1e-3


def test_handoff_reads_a_directory_written_the_reference_way(tmp_path):
    """Row f4: the experiment directory as ``FaceSwapVideoPipeline`` writes it — ``targets[i].save(imgs/T_%04d.png)``,
    ``Image.fromarray(T_mask[i]).save(mask/T_mask_%04d.png)`` (face_swap_video_pipeline.py:226-229), ``torch.save(driven_style_vector,
    styleVec/D_style_vec_%04d.pt)`` with ``[1, 12, 1280]`` tensors (:351-354), 512-pixel crops that the reader resizes to 1024 like
    ``Image.open(...).convert("RGB").resize((1024, 1024))`` (:408-411) — goes through ``handoff.load`` into the tensors the engine takes,
    and the image normalisation equals the oracle's ToTensor + Normalize restatement."""
    import numpy as np
    import torch
    from PIL import Image
    from e4s2024_amd import handoff
    from oracle import e4s_oracle as O
    rs = np.random.RandomState(1)
    n = 2
    imgs = {t: [Image.fromarray(rs.randint(0, 256, (512, 512, 3)).astype(np.uint8)) for _ in range(n)] for t in "DT"}
    masks = {t: [rs.randint(0, 12, (512, 512)).astype(np.uint8) for _ in range(n)] for t in "DT"}
    vecs = {t: [torch.from_numpy(rs.standard_normal((1, 12, 1280)).astype(np.float32)) for _ in range(n)] for t in "DT"}
    for sub in ("imgs", "mask", "styleVec"):
        os.makedirs(tmp_path / sub)
    for i in range(n):
        for t in "DT":
            imgs[t][i].save(os.path.join(tmp_path, "imgs", "%s_%04d.png" % (t, i)))
            Image.fromarray(masks[t][i]).save(os.path.join(tmp_path, "mask", "%s_mask_%04d.png" % (t, i)))
            torch.save(vecs[t][i], os.path.join(tmp_path, "styleVec", "%s_style_vec_%04d.pt" % (t, i)))
    Image.fromarray(masks["D"][0]).save(os.path.join(tmp_path, "mask", "S_mask.png"))           # files the reader must ignore
    Image.fromarray(rs.randint(0, 256, (512, 512, 3)).astype(np.uint8)).save(os.path.join(tmp_path, "mask", "D_mask_vis_0000.png"))
    clip = handoff.load(str(tmp_path))
    assert len(clip) == n and tuple(clip.target.shape) == (n, 3, 1024, 1024) and clip.target_mask.dtype == torch.uint8
    for k, t in (("driven", "D"), ("target", "T")):
        want_u8 = np.stack([np.asarray(Image.open(os.path.join(tmp_path, "imgs", "%s_%04d.png" % (t, i))).convert("RGB").resize((1024, 1024))) for i in range(n)])
        assert torch.equal(getattr(clip, k), O.frames_to_tensor(want_u8))
        assert np.array_equal(getattr(clip, k + "_mask").numpy(), np.stack(masks[t]))
        assert torch.equal(getattr(clip, k + "_style"), torch.cat(vecs[t]))
    assert len(handoff.load(str(tmp_path), first_index=1)) == 1 and len(handoff.load(str(tmp_path), count=1)) == 1


def test_bench_spawns_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` outside a launcher starts torch.distributed.run with N ranks of itself as a CHILD process (never an exec of
    a process that may have touched the GPU) and exits with its code; inside a launcher (WORLD_SIZE set) it does not spawn."""
    sys.path.insert(0, ROOT)
    import subprocess as sp
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(sp, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as ex:
        bench._spawn_ranks(4)
    assert ex.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "5", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def test_stream_pipeline_with_one_stream_is_a_plain_call():
    """``runner.StreamPipeline(1)`` needs no device: ``submit`` is the call itself, in order."""
    from e4s2024_amd.runner import StreamPipeline
    seen = []
    with StreamPipeline(1) as sp:
        out = [sp.submit(lambda a, b=0: seen.append((a, b)) or a + b, i, b=10) for i in range(4)]
    assert out == [10, 11, 12, 13] and seen == [(0, 10), (1, 10), (2, 10), (3, 10)]


def test_invalidate_weight_caches_reaches_nested_containers():
    """``bottleneck_IR_SE_Ours._wino`` is a list of tuples (direct-conv cache, Winograd cache, DMA-fed kernels' slots); a write behind
    autograd's back followed by ``invalidate_weight_caches`` must reset all of them, or the Winograd route keeps the old weights."""
    from e4s2024_amd import ops
    from e4s2024_amd.dropin.models.encoders.psp_encoders import bottleneck_IR_SE_Ours
    unit = bottleneck_IR_SE_Ours(64, 64, 1)
    caches = list(unit._w) + [c for tup in unit._wino for c in tup[1:]]
    assert len({id(c) for c in caches}) == 7          # 3 direct + 2 x (Winograd, mx row slots)
    for c in caches:
        c._state = (("stale",), None, None, None, frozenset())
        assert c.key == ("stale",)
    assert ops.invalidate_weight_caches(unit) == 7          # every distinct cache once, however many containers name it
    assert all(c.key is None for c in caches)


def test_chain_supported_knows_the_last_layer():
    """``e4s_chain_conv3x3`` exists as 64 -> 64 with a split-plane hand-over and as 32 -> 32 without one; a 64-channel LAST layer
    (``Generator(512)``) must not be routed to the chain."""
    from e4s2024_amd import ops
    assert ops.chain_supported(64, 64, 512, 512, False) and not ops.chain_supported(64, 64, 512, 512, False, last=True)
    assert ops.chain_supported(32, 32, 1024, 1024, False, last=True) and not ops.chain_supported(32, 32, 1024, 1024, False)
