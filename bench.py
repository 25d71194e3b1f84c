#!/usr/bin/env python3
"""Headline benchmark: 1024x1024 faces/sec of the region-aware StyleGAN2 synthesis (BASELINE.json configs[1]:
``Net3.gen_img`` from random W+ codes and random 12-class masks, batch 4 per GPU, randomize_noise=False).

    python bench.py --gpus N --steps K --warmup W        (N>1: one rank per GPU — launched by torch.distributed.run, or, when called
                                                          plainly, bench.py starts that launcher itself as a child process)

A step = one ``gen_img`` call on a batch of 4 faces whose codes / masks / weights are already resident in HBM; every step gets a fresh
one-hot mask tensor object, so the per-frame mask -> region-map conversion runs inside the timed region (no cache hit).
Frames are independent units: with N GPUs every rank synthesises its own batch (weak scaling, no data-path collective;
RCCL is used for the start/stop barriers and the max-over-ranks reduction of the elapsed time only).

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     — dominant kernel (the split-bf16 MFMA implicit-GEMM modulated conv): algorithmic FLOPs of its launches in a step
                 / their HIP-event durations measured inside the timed region, against dense bf16 MFMA peak / 3 MFMAs per product
  cpu_baseline — the faithful 12-pass CPU oracle timed on one face on this host's cores (rank 0, N=1 only)
  full_swap    — BASELINE configs[2] (N=1): p50 ms/frame at batch 8, its own roofline fraction and a parity check of one face of the
                 timed batch against the CPU oracle chain
  clip         — BASELINE configs[4]: a 256-frame clip, frames block-sharded over the ranks, broadcast of the clip's shared W+
                 (latent_avg) and the streamed uint8 gather of the frames to rank 0 INSIDE the timed region (every N)
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FP32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (the 5 PF marketing figure includes 2:1 sparsity)
BATCH = 4
SWAP_BATCH = 8                    # BASELINE configs[2]: full swap at batch 8


# tools/probes/taploop_probe.hip on MI355X: the dominant kernel's tap loop in isolation (same LDS layout, prefetch, barriers) sustains
# 1255 TFLOP/s of bf16 MFMA work on random operands (1530 on constant operands; board power, not issue slots, is the limit)
SUSTAINED_BF16_TFLOPS_RANDOM_DATA = 1255.0


def mfma_cost_per_product(kernel: str) -> float:
    """Nominal matrix-pipe time per algorithmic multiply-add of a kernel's arithmetic, in units of one bf16 MFMA multiply-add:
    split-bf16 = 3 bf16 MFMAs; the mx kernel's f16 + 2 x MX-fp6 = one f16 MFMA (bf16 rate) + two fp6 MFMAs at 4x the rate whose K = 64 holds the
    24 products of a kernel row x 8 channels (8 slots idle): 1 + 2 * (1/4) * (32/24) = 1.667."""
    if kernel.startswith("region_modconv_mx_kernel<1"):
        return 1.0 + 2.0 * 0.25 * 32.0 / 24.0
    if kernel.startswith("conv3x3_mx3"):      # csrc/conv_mx3.hip: the fp6 MFMAs' K = 64 holds two taps x 32 channels, nine of a chunk's ten tap slots are used
        return 1.0 + 2.0 * 0.25 * 10.0 / 9.0
    return 3.0


def conv3x3_flops_per_face(size=1024, want_executed=False, uniform_frac=None):
    """Algorithmic FLOPs of the 3x3 modulated convs per face, each counted once, keyed by the kernel that runs them
    (the transposed convs are counted per INPUT pixel).  Layers up to 256x256 are masked (12 regions).
    ``uniform_frac[out_res]``: share of a masked up layer's 16 x 16 output blocks that lie under one region — those run in the block kernel
    (csrc/modconv_upblock.hip, 2.0x their algorithmic MACs), the rest in the composed form (4x)."""
    from e4s2024_amd import ops as _ops
    from e4s2024_amd.ops import modconv_kernel_name
    ch = {4: 512, 8: 512, 16: 512, 32: 512, 64: 512, 128: 256, 256: 128, 512: 64, 1024: 32}
    out = {}
    executed = {}

    def add(cout, w_in, fl, out_res, up):
        masked = out_res <= 256
        two_stage = up and not masked and _ops.MODCONV_MODE == "sb" and _ops.UP_TWO_STAGE
        hc = two_stage and _ops.UP_FUSED and _ops.UP_HC and _ops.SP_CHAIN and cout % 32 == 0       # the chain's up layers: half-composed form (csrc/modconv_uphc.hip)
        if hc:
            k = "modconv_up_hc"
        elif two_stage:
            k = "modconv_up_fused_sb" if _ops.UP_FUSED else "modconv_tconv_sb"
        elif not masked and not up and _ops.SP_CHAIN and _ops.NHWC_CHAIN and _ops.FUSE_RGB and _ops.chain_supported(cout, cout, out_res, out_res, False, last=(out_res == size)):
            k = f"chain_conv3x3<{cout}>"          # the split-plane chain's persistent kernel (csrc/modconv_chain.hip)
        else:
            k = modconv_kernel_name(cout, w_in, None, masked, cin=None)
        f = 0.0
        if up and masked and uniform_frac and _ops.UP_BLOCKS and _ops.MODCONV_MODE == "sb" and w_in >= max(32, _ops.UP_BLOCKS_MIN_WIDTH) and cout >= 128:
            f1, f2 = uniform_frac.get(out_res, (0.0, 0.0))
            f = f1 + f2
            out["masked_upconv_blocks"] = out.get("masked_upconv_blocks", 0.0) + fl * f
            executed["masked_upconv_blocks"] = executed.get("masked_upconv_blocks", 0.0) + fl * (f1 * 2.0 + f2 * 2.5)
        out[k] = out.get(k, 0.0) + fl * (1.0 - f)
        # MACs the kernel really executes: the parity-composed up-conv spends 4x the transposed conv's, the fused one 1.31x (tile overlap)
        # (half-composed: 2x for the vertical blur factor in the weights, x 16 / 14 for the horizontal tile overlap)
        executed[k] = executed.get(k, 0.0) + fl * (1.0 - f) * (2.0 * 16 / 14 if hc else (1.31 if _ops.UP_FUSED else 1.0) if two_stage else (4.0 if up else 1.0))
    add(512, 4, 2 * 512 * 512 * 9 * 16, 4, False)
    cin, r = 512, 8
    while r <= size:
        co = ch[r]
        add(co, r // 2, 2 * cin * co * 9 * (r // 2) ** 2, r, True)      # up-conv: launched on the input grid
        add(co, r, 2 * co * co * 9 * r * r, r, False)
        cin, r = co, r * 2
    if want_executed:
        return out, executed
    return out


def _by_layer(kt, kernel, bs, peak, uniform_frac):
    """Per-layer split of the dominant kernel's launches (plus, for the masked up layers, their region-uniform blocks in the block kernel): ms per
    layer and step, algorithmic TFLOP/s and fraction of `peak`, executed / algorithmic."""
    per = {}
    for name in (kernel, "masked_upconv_blocks"):
        for detail, (calls, ms) in kt.by_detail(name).items():
            c, t = per.get(detail, (0, 0.0))
            per[detail] = (max(c, calls), t + ms)
    rows = []
    for detail, (calls, ms) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        chans, res = detail.split(" @")
        cin, cout = (int(v) for v in chans.split("->"))
        up = res.endswith(" up")
        h = int(res.split()[0])
        gflop = 2.0 * cin * cout * 9 * h * h * bs / 1e9            # per launch; up layers counted on the input grid (transposed conv)
        t = ms / calls
        from e4s2024_amd import ops as _o
        f1, f2 = uniform_frac.get(2 * h, (0.0, 0.0)) if (up and uniform_frac and h >= max(32, _o.UP_BLOCKS_MIN_WIDTH) and cout >= 128) else (0.0, 0.0)
        rows.append({"layer": detail, "ms_per_step": round(t, 4), "algorithmic_tflops": round(gflop / t, 1), "frac": round(gflop / t / peak, 4),
                     "executed_over_algorithmic": round(f1 * 2.0 + f2 * 2.5 + (1.0 - f1 - f2) * 4.0, 2) if up else 1.0,
                     **({"uniform_block_share": round(f1, 3), "uniform_sub_block_share": round(f2, 3)} if up else {})})
    return rows


TRAFFIC_FILE = "profiles/r06_traffic.json"


def _pmc_traffic(kernel_key):
    """(HBM bytes per launch of the dominant kernel, where that number comes from).  bench.py cannot collect PMC counters itself — a
    rocprofv3 --pmc run is a separate process around this very command (tools/final_prof.sh: FETCH_SIZE and WRITE_SIZE in separate passes,
    written to TRAFFIC_FILE) — so the figure is NOT live: it is reported only if the committed passes of THIS round cover the kernel that
    dominates now, and tagged with its source; otherwise null."""
    path = os.path.join(ROOT, TRAFFIC_FILE)
    try:
        with open(path) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None, f"none: {TRAFFIC_FILE} not present"
    ent = t.get(kernel_key)
    if ent is None:
        return None, f"none: {TRAFFIC_FILE} has no pass for {kernel_key}"
    return ent.get("hbm_bytes_per_launch"), f"committed rocprofv3 --pmc passes of this command ({TRAFFIC_FILE}; not measured in this run)"


# BASELINE configs[2] unit of work: 2 x BiSeNet (26.77) + 2 x encoder (229.34) + MLPs (0.10) + synthesis (148.52) GFLOP per face
FULL_SWAP_GFLOP = {"parser": 2 * 26.77, "encoder": 2 * 229.34, "mlps": 0.10, "synthesis": 148.52}


def _spawn_ranks(n):
    """``python bench.py --gpus N`` outside a launcher: start ``torch.distributed.run`` with N ranks of this script as a CHILD process and
    exit with its code.  Nothing in this process has touched the GPU yet (and nothing will)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def _overlap_view(kt, dom, per_launch_flops, peak):
    """The dominant kernel's launches inside the overlapped timed region: average duration between their own events and what it prices to."""
    if kt is None or dom is None:
        return None
    s = kt.summary().get(dom)
    if not s or not s[0]:
        return None
    avg_ms = s[1] / s[0]
    ach = per_launch_flops / (avg_ms * 1e-3) / 1e12
    return {"avg_launch_ms": round(avg_ms, 4), "achieved": round(ach, 2), "frac": round(ach / peak, 4), "launches": s[0]}


class _BoardSampler:
    """Package power and shader clock while a block runs, read from `rocm-smi --showpower --showclocks` four times a second on a host thread (no GPU work, no
    root).  Round 6 finding (tools/power_probe.sh, tools/power_by_kernel.py): every kernel of the step runs at the board's power limit and the shader clock
    settles near 2.0 GHz instead of 2.4 — the ceiling the roofline fractions are quoted against (2 500 TFLOP/s at 2.4 GHz) is not reachable on this workload."""

    def __init__(self):
        self.samples = []
        self._stop = None
        self._th = None

    def _run(self):
        import re
        import subprocess
        while not self._stop.is_set():
            try:
                txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
                pw = re.search(r"Power \(W\): ([0-9.]+)", txt)
                ck = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
                if pw and ck:
                    self.samples.append((time.perf_counter(), float(pw.group(1)), int(ck.group(1))))
            except Exception:      # noqa: BLE001 - a side measurement
                return
            self._stop.wait(0.25)

    def __enter__(self):
        import threading
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)
        self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._th.join(timeout=10)

    def summary(self, t0):
        """Mean over the samples taken later than 0.5 s after ``t0`` (the settled part)."""
        s = [x for x in self.samples if x[0] >= t0 + 0.5] or self.samples
        if not s:
            return None
        return {"power_w": round(sum(x[1] for x in s) / len(s), 0), "sclk_mhz": round(sum(x[2] for x in s) / len(s), 0), "samples": len(s),
                "what": "rocm-smi package power / shader clock during the soak (the timed step repeated): the board's power limit, not issue slots, sets the clock"}


def _compact_line(line, one_stream, ksum, steps, detail_path):
    """The ONE stdout line: the contract fields, `roofline` / `cpu_baseline` with scalars only, and the secondary metrics as flat scalars (< 2 000 characters)."""
    def short(v, n=120):
        return v if not isinstance(v, str) or len(v) <= n else v[:n - 1] + "~"

    def scalars(d, keep=None):
        return {k: short(v) for k, v in (d or {}).items() if not isinstance(v, (dict, list)) and (keep is None or k in keep)} or None

    def get(d, *path):
        for k in path:
            d = d.get(k) if isinstance(d, dict) else None
        return d
    out = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline")}
    out["dtype"] = short(line["dtype"], 80)
    out["data"] = line["data"]
    cfg = line["config"]
    out["config"] = {"workload": f"BASELINE configs[1]: StyleGAN2 1024x1024 synthesis from random W+, 12-region masks, batch={cfg['batch_per_gpu']}/GPU",
                     "batch_per_gpu": cfg["batch_per_gpu"], "global_batch": cfg["global_batch"], "parallelism": cfg["parallelism"], "streams_per_gpu": cfg["streams_per_gpu"]}
    roof = line.get("roofline")
    r = scalars(roof, {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches_per_step", "avg_launch_ms", "algorithmic_gflop_per_launch", "whole_job_frac"})
    if r is not None:
        r["overlapped_avg_launch_ms"] = get(roof, "in_overlapped_region", "avg_launch_ms")
    out["roofline"] = r
    out["cpu_baseline"] = scalars(line.get("cpu_baseline"), {"value", "unit", "cores", "kind", "sample", "max_abs_pixel_diff_vs_gpu"})
    if out["cpu_baseline"] and "sample" in out["cpu_baseline"]:
        out["cpu_baseline"]["sample"] = short(out["cpu_baseline"]["sample"], 56)
    # stage split of a one-stream step (ms): masked 3x3 layers 32^2..256^2, the single-region >= 512^2 stage, everything else (4^2-16^2 head, ToRGBs, tables)
    stage = None
    if ksum and one_stream:
        ms = lambda pred: sum(v[1] for k, v in ksum.items() if pred(k)) / steps    # noqa: E731
        masked = ms(lambda k: k.startswith(("region_modconv_mx_kernel", "region_modconv_sb_kernel<4,", "masked_upconv_blocks")))
        ge512 = ms(lambda k: k.startswith(("modconv_up_hc", "modconv_up_fused_sb", "chain_conv3x3", "chain_fused1024", "modconv_tconv_sb")))
        stage = {"masked": round(masked, 3), "ge512": round(ge512, 3), "rest": round(max(0.0, one_stream["ms_per_step"] - masked - ge512), 3)}
    out["stage_ms"] = stage
    out["one_stream_faces_per_s"] = get(line, "one_stream", "faces_per_s")
    out["soak_faces_per_s"] = get(line, "soak", "faces_per_s")
    out["soak_power_w"] = get(line, "soak", "board", "power_w")
    out["soak_sclk_mhz"] = get(line, "soak", "board", "sclk_mhz")
    # at the board's power limit the honest ceiling is the nominal one scaled by the clock the board grants (2 400 MHz nominal), and the figure of merit is energy per face
    pw, ck, fps = out["soak_power_w"], out["soak_sclk_mhz"], out["soak_faces_per_s"]
    if r is not None and ck:
        r["frac_at_soak_sclk"] = round(r["frac"] * 2400.0 / ck, 4)
    out["soak_joules_per_face"] = round(pw / fps, 3) if (pw and fps) else None
    fs = line.get("full_swap") or {}
    out["full_swap_p50_ms_per_frame"] = fs.get("p50_ms_per_frame")
    out["full_swap_swaps_per_s"] = fs.get("swaps_per_s")
    out["full_swap_overlapped_ms_per_frame"] = get(fs, "overlapped_batches", "ms_per_frame")
    out["full_swap_frac"] = get(fs, "roofline", "frac")
    out["full_swap_max_abs_pixel_diff"] = get(fs, "parity", "max_abs_pixel_diff_vs_oracle")
    out["full_swap_label_flips"] = get(fs, "parity", "parser_label_flips_vs_oracle")
    pt = line.get("pti") or {}
    out["pti_s_per_iter"] = pt.get("s_per_iter", short(pt.get("error"), 60) if pt else None)
    out["pti_frac"] = get(pt, "roofline", "frac")
    out["clip_frames_per_s"] = get(line, "clip", "frames_per_s")
    out["clip_ms_per_frame"] = get(line, "clip", "ms_per_frame")
    msn = line.get("mask_sensitivity") or {}
    out["mask_faces_per_s"] = {k.split("_")[0]: v.get("faces_per_s") for k, v in msn.items()} or None
    out["f16_overflowed"] = get(line, "f16_range", "overflowed_in_the_measured_passes")
    out["detail"] = "stderr" + (", gpurun_out/bench_detail.json" if detail_path else "")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--streams", type=int, default=2, help="HIP streams consecutive steps alternate over (1 = one stream)")
    ap.add_argument("--no-mask-sensitivity", action="store_true", help="skip the two short runs under coarse / i.i.d. region maps")
    ap.add_argument("--labels", choices=["blocky", "coarse", "portrait", "iid"], default="blocky",
                    help="region maps: 16 x 16 constant cells on the 512 x 512 map (default, BASELINE configs[1]), 4 x 4 cells (face-sized regions), or i.i.d. per pixel")
    ap.add_argument("--soak-seconds", type=float, default=2.0, help="after the timed K steps: the same steps for at least this long (sustained rate, a side field); 0 skips it")
    ap.add_argument("--timed-events", choices=["dominant", "none"], default="none",
                    help="HIP events inside the timed region: on the dominant kernel's launches (roofline.in_overlapped_region comes from the timed steps themselves) or none "
                         "(that view then comes from an instrumented repeat of the same K steps right behind the timed region)")
    ap.add_argument("--settle-seconds", type=float, default=0.0, help="experiment: untimed steps for this long in front of the warm-up (clock / power settling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-in-run-ab", action="store_true", help="skip the extra one-stream pass on the previous rounds' kernels (roofline.in_run_ab); the profiling scripts use it "
                                                                   "so that their kernel tables list the default routes only")
    ap.add_argument("--no-full-swap", action="store_true", help="skip the secondary full-swap p50 measurement")
    ap.add_argument("--no-pti", action="store_true", help="skip the secondary PTI step measurement (BASELINE configs[3])")
    ap.add_argument("--pti-passes", type=int, default=5, help="passes of the PTI clip loop over its 32 frames (BASELINE configs[3] states 200: ~80 s on one GPU; default 5)")
    ap.add_argument("--clip", type=int, default=256, help="frames of the clip-mode measurement (BASELINE configs[4]); 0 skips it")
    ap.add_argument("--clip-batch", type=int, default=SWAP_BATCH)
    ap.add_argument("--clip-unit", choices=["swap", "gen"], default="swap",
                    help="per-frame work of the clip: the reference's per-frame swap (2 parses + 2 encodes + mask surgery + mix + synthesis) or synthesis only")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _spawn_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm

    import e4s2024_amd
    from e4s2024_amd import ops, seeded
    e4s2024_amd.install()
    from models.networks import Net3
    import argparse as _ap

    opts = _ap.Namespace(fsencoder_type="psp", remaining_layer_idx=13, num_seg_cls=12, out_size=1024, train_G=False,
                         start_from_latent_avg=True, learn_in_w=False)
    net = Net3(opts).eval()
    seeded.apply_seeded(net.G, 4, "net3", prefix="G.")          # synthesis only: encoder / MLP weights are not touched by gen_img
    la = seeded.seeded_latent_avg(2, 18)
    net.latent_avg = la.to(dev)
    sd_cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sd_cpu = {"G." + k: v.clone() for k, v in net.G.state_dict().items()}
    net = net.to(dev)

    bs = args.batch
    # codes = latent_avg + 0.5 N(0,1) (seed 1 + rank), blocky 16x16-cell label maps (seed 3 + rank)
    codes = seeded.seeded_codes(1 + rank, bs, 12, 18, la).to(dev)
    lab = (seeded.facelike_labels(5 + rank, bs, 512) if args.labels == "portrait" else
           seeded.blocky_labels(3 + rank, bs, 12, 512, 16 if args.labels == "blocky" else 4) if args.labels != "iid" else seeded.iid_labels(9 + rank, bs, 12, 512))
    mask = seeded.labels_to_onehot(lab, 12).to(dev)
    ops.STRICT_MASK = False                                       # the one-hot check costs a host sync; masks here are one-hot by construction
    # f16 range guard (ops.MxGuard): gen_img would wait for its own guard after every pass (one host synchronisation per call); the benchmark owns ONE guard
    # over all its synthesis passes instead and reports it — a value measured on passes that left the f16 range would be a value of broken frames
    _guard_scope = ops.mx_guard_scope()
    guard_all = _guard_scope.__enter__()

    def step():
        # a fresh tensor OBJECT per step (a view: no copy, no extra kernel): ops.mask_to_labels caches per mask object, and a frame's
        # one-hot mask -> uint8 region map conversion is per-frame work that belongs inside the timed region
        with torch.no_grad():
            return net.gen_img(None, codes, mask.view_as(mask), randomize_noise=False)[0]

    # Steps are independent batches; consecutive ones go to alternating HIP streams (runner.StreamPipeline) so that the latency-bound 4^2-32^2
    # layers of one batch run under the large layers of the batch before.  Exactly K steps between the two synchronisation points, every one
    # complete at the second; --streams 1 = one stream.
    from e4s2024_amd.runner import StreamPipeline
    pipe = StreamPipeline(args.streams, device=dev)
    with ops.KernelTimer() as k0:                    # which kernel dominates (one untimed step): only its launches carry events in the timed region
        img = step()
    s0 = k0.summary()
    dom0 = max(s0, key=lambda k: s0[k][1]) if s0 else None
    if args.settle_seconds > 0:
        t_set = time.perf_counter()
        with pipe:
            while time.perf_counter() - t_set < args.settle_seconds:
                for _ in range(args.steps):
                    img = pipe.submit(step)
                torch.cuda.synchronize()
    with pipe:
        for _ in range(max(args.warmup, 2 * args.streams if args.warmup else 0)):      # (every stream's workspace / control words exist)
            img = pipe.submit(step)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bare = args.timed_events == "none" and args.streams > 1
    with (ops.KernelTimer(only={dom0} if (args.streams > 1 and dom0) else None) if not bare else contextlib.nullcontext()) as kt:
        with pipe:
            for _ in range(args.steps):
                img = pipe.submit(step)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    if bare:                                   # the overlapped view of the dominant kernel: the same K steps once more, instrumented
        with ops.KernelTimer(only={dom0} if dom0 else None) as kt:
            with pipe:
                for _ in range(args.steps):
                    img = pipe.submit(step)
            torch.cuda.synchronize()
    kt_overlap = kt if args.streams > 1 else None
    # sustained rate: the same step on the same streams for >= --soak-seconds (the K = 20 steps of the contract last ~60 ms, shorter than the board's
    # power / clock settling); reported beside `value`, never instead of it
    soak = None
    if args.soak_seconds > 0 and world == 1:
        n_soak = 0
        t_s0 = time.perf_counter()
        with _BoardSampler() as board, pipe:
            while True:
                for _ in range(2 * args.steps):
                    img_s = pipe.submit(step)
                n_soak += 2 * args.steps
                torch.cuda.synchronize()
                if time.perf_counter() - t_s0 >= args.soak_seconds:
                    break
        t_soak = time.perf_counter() - t_s0
        soak = {"seconds": round(t_soak, 3), "steps": n_soak, "faces_per_s": round(n_soak * bs / t_soak, 1), "ms_per_step": round(t_soak / n_soak * 1e3, 3),
                "streams": args.streams, "what": "the timed region's step repeated for at least --soak-seconds (synchronised every 2K steps)",
                "board": board.summary(t_s0)}
        del img_s
    one_stream = None
    if args.streams > 1:
        # the same K steps on ONE stream, twice: (a) with events on the dominant kernel's launches only, exactly the instrumentation of the timed region —
        # the one-stream HEADLINE (comparable with `value` and with earlier rounds' one-stream figures); (b) with every launch bracketed by events (~4 % of
        # a step): a kernel's duration between its own events is its rate only while it has the chip to itself (under the overlap two batches' kernels
        # share it), so the roofline object is computed from pass (b)
        for _ in range(2):
            img1 = step()
        torch.cuda.synchronize()
        t1s = time.perf_counter()
        with ops.KernelTimer(only={dom0} if dom0 else None):
            for _ in range(args.steps):
                img1 = step()
            torch.cuda.synchronize()
            e1a = time.perf_counter() - t1s
        t1s = time.perf_counter()
        with ops.KernelTimer() as kt:
            for _ in range(args.steps):
                img1 = step()
            torch.cuda.synchronize()
            e1 = time.perf_counter() - t1s
        one_stream = {"faces_per_s": round(args.steps * bs / e1a, 1), "ms_per_step": round(e1a / args.steps * 1e3, 3),
                      "faces_per_s_every_launch_timed": round(args.steps * bs / e1, 1),
                      "images_equal_overlapped": bool(torch.equal(img1, img)),
                      "what": "the same K steps on one stream: `faces_per_s` with events on the dominant kernel's launches only (comparable with earlier rounds' one-stream figures), "
                              "`faces_per_s_every_launch_timed` with every instrumented launch bracketed by HIP events (the pass `roofline` is computed from)"}
        del img1
    ksum = kt.summary()
    kt_for_layers = kt
    # In-run A/B against the previous rounds' kernels (boxes of the pool differ by 7-15 % on untouched kernels, so `value` alone cannot show a 5 % gain):
    # the same K steps on one stream, every launch timed, with the routes switched back — masked 3x3 layers on the register-staged split-bf16 kernel of
    # round 2 (E4S_MX=0), the chain's up layers on round 3's fused LDS-DMA kernel (E4S_UP_HC=0).  Outside the timed region.
    in_run_ab = None
    if world == 1 and ops.MODCONV_MODE == "sb" and ksum and not args.no_in_run_ab:
        saved_routes = (ops.MX_MODE, ops.UP_HC)
        try:
            ops.MX_MODE, ops.UP_HC = 0, False
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            with ops.KernelTimer() as kb:
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
        finally:
            ops.MX_MODE, ops.UP_HC = saved_routes
        step()                                                        # (back on the default routes: weight copies rebuilt outside anything timed)
        torch.cuda.synchronize()
        base = kb.summary()
        groups = {"masked_3x3_layers_32_to_256": lambda k: k.startswith(("region_modconv_mx_kernel", "region_modconv_sb_kernel<4,", "masked_upconv_blocks")),
                  "single_region_up_layers_256_to_1024": lambda k: k.startswith(("modconv_up_hc", "modconv_up_fused_sb")),
                  "single_region_convs_512_1024": lambda k: k.startswith("chain_conv3x3")}
        in_run_ab = {"what": "ms per step of a kernel group on one stream in THIS run: `ms_prev` with the previous rounds' routes (E4S_MX=0: round 2's masked kernel; "
                             "E4S_UP_HC=0: round 3's fused up kernel), `ms_now` with the default routes"}
        for gname, pred in groups.items():
            prev = sum(v[1] for k, v in base.items() if pred(k)) / args.steps
            now = sum(v[1] for k, v in ksum.items() if pred(k)) / args.steps
            in_run_ab[gname] = {"ms_prev": round(prev, 4), "ms_now": round(now, 4), "ratio": round(now / prev, 4) if prev > 0 else None}
        in_run_ab["all_launches"] = {"ms_prev": round(sum(v[1] for v in base.values()) / args.steps, 4), "ms_now": round(sum(v[1] for v in ksum.values()) / args.steps, 4)}
        # ... and the masked up layers with / without the four-parity kernel of round 4 (csrc/modconv_mx4.hip; everything else on the default routes)
        if ops.UP_MX4 and dom0:
            try:
                ops.UP_MX4 = False
                step()
                torch.cuda.synchronize()
                with ops.KernelTimer() as k4:
                    for _ in range(args.steps):
                        step()
                    torch.cuda.synchronize()
            finally:
                ops.UP_MX4 = True
            prev4, now4 = k4.by_detail(dom0), kt_for_layers.by_detail(dom0)
            in_run_ab["masked_up_layers_four_parity_kernel"] = {
                d: {"ms_composed_kernel_alone": round(prev4[d][1] / prev4[d][0], 4), "ms_now": round(now4[d][1] / now4[d][0], 4)}
                for d in sorted(now4) if d.endswith(" up") and d in prev4 and now4[d][0] and prev4[d][0]}
    # how much of `value` depends on the region maps: the same batch under portrait-shaped maps (ellipses: hair, skin, eyes, ...: what the face
    # parser produces on photographs), under 4 x 4 cells (every 16 x 16 block of the masked up layers lies under one region) and under i.i.d.
    # per-pixel labels (none does); single-GPU runs only, 10 steps each, outside the timed region
    mask_sens = None
    if world == 1 and args.labels == "blocky" and not args.no_mask_sensitivity:
        mask_sens = {}
        for name, lb in (("portrait_like_ellipses", seeded.facelike_labels(5, bs, 512)), ("coarse_4x4_cells", seeded.blocky_labels(3, bs, 12, 512, 4)),
                         ("iid_per_pixel", seeded.iid_labels(9, bs, 12, 512))):
            m2 = seeded.labels_to_onehot(lb, 12).to(dev)
            with torch.no_grad():
                with pipe:
                    for _ in range(2 * args.streams):
                        pipe.submit(net.gen_img, None, codes, m2.view_as(m2), randomize_noise=False)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                with pipe:                                               # (the same step overlap as the timed region)
                    for _ in range(10):
                        pipe.submit(net.gen_img, None, codes, m2.view_as(m2), randomize_noise=False)
                torch.cuda.synchronize()
            mask_sens[name] = {"faces_per_s": round(10 * bs / (time.perf_counter() - t1), 1)}
            del m2
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(img).all()
    guard_all.arm()
    _guard_scope.__exit__(None, None, None)
    f16_overflowed = guard_all.tripped()

    # ---- the full-swap models (encoder, per-region MLPs, parser): seeded on every rank, used by the full-swap and clip measurements
    parser = None
    need_swap_models = (rank == 0 and world == 1 and not args.no_full_swap) or (args.clip > 0 and args.clip_unit == "swap")
    if need_swap_models:
        from e4s2024_amd import pipeline
        from swap_face_fine.face_parsing.face_parsing_demo import FaceParser
        seeded.apply_seeded(net.encoder, 4, "net3", prefix="encoder.")
        for i, m in enumerate(net.MLPs):
            seeded.apply_seeded(m, 4, "net3", prefix=f"MLPs.{i}.")
        parser = FaceParser(None, device=dev)
        seeded.apply_seeded(parser.seg, 7, "bisenet")
        parser.seg.eval()

    # ---- secondary metric of BASELINE.json: p50 ms/frame of the full swap (2 parses + 2 encodes + MLPs + synthesis), batch 8
    full_swap = None
    if rank == 0 and world == 1 and not args.no_full_swap:
        drv = seeded.seeded_image(5, SWAP_BATCH, 1024).to(dev)
        tgt = seeded.seeded_image(6, SWAP_BATCH, 1024).to(dev)
        for _ in range(2):
            pipeline.swap_batch(net, parser, drv, tgt)
        torch.cuda.synchronize()
        times = []
        for _ in range(13):                                   # 13 x 8 = 104 frames
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); frames, labs = pipeline.swap_batch(net, parser, drv, tgt); b.record()
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b))
        times.sort()
        p50 = times[len(times) // 2]
        # throughput of consecutive batches on alternating streams (what the clip loop does: runner.run_clip_streamed); the p50 above is the
        # latency of one batch that has the chip to itself
        swap_overlapped = None
        if args.streams > 1:
            with pipe:
                for _ in range(args.streams):
                    pipe.submit(pipeline.swap_batch, net, parser, drv, tgt)
            torch.cuda.synchronize()
            t_sw = time.perf_counter()
            swap_guards = []                          # (batches in flight: each call's guard is looked at after the loop instead of awaited inside it)
            with pipe:
                for _ in range(12):
                    fr2 = pipe.submit(pipeline.swap_batch, net, parser, drv, tgt, guard=swap_guards)[0]
            torch.cuda.synchronize()
            t_sw = time.perf_counter() - t_sw
            f16_overflowed = f16_overflowed or any(g.tripped() for g in swap_guards)
            swap_overlapped = {"swaps_per_s": round(12 * SWAP_BATCH / t_sw, 1), "ms_per_frame": round(t_sw / (12 * SWAP_BATCH) * 1e3, 3), "batches": 12,
                               "streams": args.streams, "frames_equal_one_stream": bool(torch.equal(fr2, frames))}
            del fr2
        # latency of ONE swap (a batch of one: what an interactive caller sees; the encoder's 512-channel convolutions take the Winograd route there)
        one_swap_ms = None
        try:
            d1, t1_ = drv[:1].contiguous(), tgt[:1].contiguous()
            for _ in range(3):
                pipeline.swap_batch(net, parser, d1, t1_)
            torch.cuda.synchronize()
            ts = []
            for _ in range(15):
                a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
                a.record(); pipeline.swap_batch(net, parser, d1, t1_); b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            one_swap_ms = round(sorted(ts)[len(ts) // 2], 3)
        except Exception as e:      # noqa: BLE001 - secondary measurement
            one_swap_ms = f"{type(e).__name__}: {e}"[:200]
        # roofline of the unit: algorithmic GFLOP per face against the bf16 MFMA peak divided by the MFMAs each part spends per
        # product (parser: three-way split = 6, everything else: two-way split = 3)
        gf = FULL_SWAP_GFLOP
        total_gf = sum(gf.values())
        # bf16-MFMA multiply-adds of matrix-pipe time per algorithmic multiply-add, part by part: the parser's fp32-class convolutions 3 (two-term
        # f16 split; 6 with the three-way bf16 split, 16 exact), the encoder 3 (split-bf16) or — its stride-1 3x3 convolutions with >= 128 output channels, 90 %
        # of its FLOPs, on the mx kernels at this batch — 1.556 (two-phase kernel) / 1.667, the synthesis as in the headline's whole-job figure
        mx = ops.mx_arith() == 1
        parser_cost = {"f16x3": 3.0, "sb3": 6.0, True: 16.0, False: 3.0}.get(ops.PARSER_EXACT, 3.0)
        enc_mx = "conv3x3_mx3_kernel" if getattr(ops, "MX3", False) else "region_modconv_mx_kernel<1>"     # (all of these layers have cin % 32 == 0)
        enc_cost = (0.9 * mfma_cost_per_product(enc_mx) + 0.1 * 3.0) if (mx and ops.CONV_MODE == "sb") else 3.0
        fl_s = conv3x3_flops_per_face()
        syn_cost = sum(fl_s[k] * mfma_cost_per_product(k) for k in fl_s) / sum(fl_s.values())
        mfma_per_product = (gf["parser"] * parser_cost + gf["encoder"] * enc_cost + gf["mlps"] * 3.0 + gf["synthesis"] * syn_cost) / total_gf
        fs_peak = BF16_MATRIX_PEAK_TFLOPS / mfma_per_product
        fs_ach = total_gf * 1e9 * SWAP_BATCH / (p50 * 1e-3) / 1e12
        full_swap = {"p50_ms_per_frame": round(p50 / SWAP_BATCH, 3), "p50_ms_per_batch": round(p50, 3), "batch": SWAP_BATCH, "frames": 13 * SWAP_BATCH,
                     "swaps_per_s": round(SWAP_BATCH / p50 * 1e3, 1), "overlapped_batches": swap_overlapped, "p50_ms_one_swap_alone": one_swap_ms,
                     "roofline": {"bound": "mfma", "achieved": round(fs_ach, 2), "peak": round(fs_peak, 1), "unit": "TFLOP/s", "frac": round(fs_ach / fs_peak, 4),
                                  "algorithmic_gflop_per_face": round(total_gf, 2),
                                  "frac_on_round2_basis": round(fs_ach / (BF16_MATRIX_PEAK_TFLOPS / ((gf["parser"] * 6 + (total_gf - gf["parser"]) * 3) / total_gf)), 4),
                                  "peak_basis": f"dense bf16 MFMA 2500 TFLOP/s / {mfma_per_product:.3f} bf16-MFMA multiply-adds of matrix-pipe time per product (parser {parser_cost:g}, "
                                                f"encoder {enc_cost:.3f}, MLPs 3, synthesis {syn_cost:.3f}); frac_on_round2_basis: parser 6, everything else 3"},
                     "unit_of_work": "2 x BiSeNet parse (fp32-class split arithmetic: E4S_PARSER_CONV) + 2 x get_style_vectors + style mix + cal_style_codes + gen_img + tensor2im, "
                                     "inputs resident in HBM (BASELINE configs[2])"}
        if sd_cpu is not None:
            # parity of the timed batch: one face of it through the CPU oracle chain (parse x2 -> style vectors x2 -> mix -> codes -> synthesis)
            try:
                from oracle import e4s_oracle as O
                import numpy as np
                torch.set_num_threads(min(16, os.cpu_count() or 1))
                fb = 3
                sd_all = {k: v.detach().cpu() for k, v in net.state_dict().items()}
                sd_bis = {k: v.detach().cpu() for k, v in parser.seg.state_dict().items()}
                with torch.no_grad():
                    lab_d_gpu = parser.parse_batch((drv[fb:fb + 1] + 1) / 2, seg12=True)[0].cpu().numpy()
                    lab_t_gpu = labs[fb].cpu().numpy()
                    flips = 0
                    for img_, got in ((drv, lab_d_gpu), (tgt, lab_t_gpu)):
                        logits = O.bisenet_forward(sd_bis, O.parser_preprocess((img_[fb:fb + 1].cpu() + 1) / 2))
                        flips += int((O.remap_19_to_12(torch.argmax(logits, 1)[0].numpy().astype(np.uint8)) != got).sum())
                    oh = lambda l: O.label_map_to_onehot(torch.from_numpy(l.astype(np.int64))[None, None], 12)   # noqa: E731
                    v_d, _ = O.get_style_vectors(sd_all, drv[fb:fb + 1].cpu(), oh(lab_d_gpu))
                    v_t, _ = O.get_style_vectors(sd_all, tgt[fb:fb + 1].cpu(), oh(lab_t_gpu))
                    mixed = O.swap_comp_style_vector(v_t, v_d, pipeline.DEFAULT_COMP_INDICES, False)
                    ref_img, _ = O.generator_forward(sd_all, O.cal_style_codes(sd_all, mixed, la, 13), oh(lab_t_gpu), None)
                    img_f, _ = pipeline.swap_batch(net, parser, drv, tgt, to_uint8=False)
                full_swap["parity"] = {"face": fb, "max_abs_pixel_diff_vs_oracle": float(f"{(img_f[fb].cpu() - ref_img[0]).abs().max().item():.3e}"),
                                       "max_grey_level_diff_vs_oracle": int(np.abs(frames[fb].cpu().numpy().astype(np.int16)
                                                                                   - O.tensor2im_array(ref_img[0]).astype(np.int16)).max()),
                                       "parser_label_flips_vs_oracle": flips, "tol": 1e-3,
                                       "how": "face 3 of the timed batch of 8 through oracle/e4s_oracle.py from the device's region maps"}
                del img_f, ref_img
            except Exception as e:      # noqa: BLE001 - a reported check must not cost the line
                full_swap["parity"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        del drv, tgt, frames, labs

    # ---- BASELINE configs[4]: a clip, frames block-sharded over the ranks (runner.FrameShardRunner); inside the timed region: the broadcast
    # of the clip's shared W+ (latent_avg, what every frame's codes are offsets of) from rank 0, each rank's per-batch swaps, and the
    # streamed uint8 gather of the finished frames to rank 0 (face_swap_video_pipeline.py:404-486 per frame)
    clip_info = None
    if args.clip > 0:
        try:
            from e4s2024_amd import runner as _runner
            rn = _runner.FrameShardRunner(device=dev)
            n_frames, cb = args.clip, args.clip_batch
            POOL = 16
            if args.clip_unit == "swap":
                pool_d = seeded.seeded_image(50, POOL, 1024).to(dev)       # resident frame pools; frame i of the clip = pool[i % 16]
                pool_t = seeded.seeded_image(60, POOL, 1024).to(dev)

                def frame_inputs(lo, hi):
                    idx = torch.arange(lo, hi, device=dev) % POOL
                    return pool_d.index_select(0, idx), pool_t.index_select(0, idx)

                def synth(shared, fi):
                    net.latent_avg = shared
                    return pipeline.swap_batch(net, parser, fi[0], fi[1], mask_surgery=True)[0]
                unit = ("per frame: 2 x parse + 2 x get_style_vectors + swap_head_mask_hole_first + style mix + cal_style_codes + gen_img + tensor2im "
                        "(pipeline.swap_batch(mask_surgery=True))")
            else:
                pool_c = seeded.seeded_codes(51, POOL, 12, 18, la).to(dev)
                pool_l = torch.from_numpy(seeded.blocky_labels(52, POOL, 12, 512, 16)).to(dev).to(torch.uint8)

                def frame_inputs(lo, hi):
                    idx = torch.arange(lo, hi, device=dev) % POOL
                    return pool_c.index_select(0, idx), pool_l.index_select(0, idx)

                def synth(shared, fi):
                    net.latent_avg = shared
                    return _runner.gen_img_frames(net, fi[0], fi[1])
                unit = "per frame: gen_img + tensor2im (runner.gen_img_frames)"
            shared_src = la.to(dev) if rank == 0 else None
            out_buf = torch.empty((n_frames, 1024, 1024, 3), dtype=torch.uint8, device=dev) if rank == 0 else None
            with torch.no_grad():
                synth(la.to(dev), frame_inputs(0, cb))                    # warm-up batch (weight caches, allocator)
            # ... and two untimed rounds through the streamed loop itself: it runs on the runner's own streams, whose 128 MB split-K workspaces, gather buffers and
            # allocator pools are first-use allocations (after the PTI section's empty_cache() they are fresh device allocations)
            rn.run_clip_streamed(min(n_frames, 2 * cb * max(1, world)), rn.broadcast_shared(shared_src, (18, 512)), frame_inputs, synth, batch=cb)
            torch.cuda.synchronize()
            rn.barrier()
            torch.cuda.synchronize()
            tc0 = time.perf_counter()
            shared = rn.broadcast_shared(shared_src, (18, 512))
            frames_all = rn.run_clip_streamed(n_frames, shared, frame_inputs, synth, batch=cb, out=out_buf)
            torch.cuda.synchronize()
            rn.barrier()
            torch.cuda.synchronize()
            clip_s = rn.max_over_ranks(time.perf_counter() - tc0)
            ok = None
            if rank == 0:
                # the last batch of the LAST rank's block, recomputed here with the same batch composition, must equal what arrived
                s_last, e_last = _runner.shard_range(n_frames, world - 1, world)
                lo = s_last + ((e_last - s_last - 1) // cb) * cb
                lo2 = max(s_last, e_last - cb) if e_last - lo < cb else lo        # (a short last round is computed on the block's last `cb` frames: runner.run_clip_streamed)
                with torch.no_grad():
                    again = synth(shared, frame_inputs(lo2, e_last))
                ok = bool(torch.equal(again[lo - lo2:], frames_all[lo:e_last]))
            # the same clip three more times back to back (`sustained_frames_per_s` = the last repetition): before round 4's runner kept its streams, every clip ran on
            # fresh streams whose allocator pools were empty, and a repetition cost 25 % more than the first run
            sustained = None
            if world == 1:
                for _ in range(3):
                    torch.cuda.synchronize()
                    ts0 = time.perf_counter()
                    rn.run_clip_streamed(n_frames, shared, frame_inputs, synth, batch=cb, out=out_buf)
                    torch.cuda.synchronize()
                    sustained = round(n_frames / (time.perf_counter() - ts0), 1)
            max_block = -(-n_frames // world)
            clip_info = {"frames": n_frames, "batch": cb, "seconds": round(clip_s, 4), "frames_per_s": round(n_frames / clip_s, 1), "sustained_frames_per_s": sustained,
                         "ms_per_frame": round(clip_s / n_frames * 1e3, 3), "scaling": "strong", "n_gpus": world, "unit_of_work": unit,
                         "collectives_in_timed_region": f"broadcast latent_avg [18,512] from rank 0 + {-(-max_block // cb)} rounds of "
                                                        f"async gather of uint8 frames ({cb} x 3 MB per rank per round) to rank 0",
                         "gathered_frames_match_recomputation": ok}
            del out_buf, frames_all
            rn.close()              # (the runner's pipelines pin two streams' contexts — 128 MB of split-K workspace each — until closed)
        except Exception as e:      # noqa: BLE001 - secondary measurement
            clip_info = {"error": f"{type(e).__name__}: {e}"[:300]}
        net.latent_avg = la.to(dev)
    del parser

    # ---- BASELINE configs[3]: one PTI optimiser step (cal_style_codes -> gen_img -> L2 -> backward -> Adam) at 1024x1024, batch 1,
    # the whole step replayed as one hipGraph.  Reported beside the headline, never part of `value`; a failure here must not cost the line.
    pti_info = None
    if rank == 0 and world == 1 and not args.no_pti:
        try:
            from e4s2024_amd import pti
            torch.cuda.synchronize()
            tnet = Net3(_ap.Namespace(**{**vars(opts), "train_G": True}))
            seeded.apply_seeded(tnet, 4, "net3")
            tnet = tnet.to(dev).train()
            tnet.latent_avg = net.latent_avg
            params = pti.trainable_parameters(tnet)
            topt = torch.optim.Adam(params, lr=1e-3, capturable=True, fused=True)
            vec = torch.from_numpy(seeded.seeded_array(41, "vec", (1, 12, 1280), dist="normal")).to(dev)
            tlab = torch.from_numpy(seeded.blocky_labels(3, 1, 12, 512, 16)).to(dev).to(torch.uint8)
            target = torch.tanh(torch.from_numpy(seeded.seeded_array(5, "img", (1, 3, 1024, 1024), dist="normal"))).to(dev)
            step = pti.GraphedPTIStep(tnet, topt, vec, tlab, target)
            l0 = step(vec, tlab, target)[0].item()
            for _ in range(2):                                # (the first replays of a new graph have run up to 10 % slow: one line of round 4 said 13.6 ms where
                step(vec, tlab, target)                       # three repeats of the section said 12.4 — the figure is now the median of three timed groups, all three reported)
            n_it, groups = 10, []
            for _ in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(n_it):
                    lN, _ = step(vec, tlab, target)
                torch.cuda.synchronize()
                groups.append((time.perf_counter() - t1) / n_it)
            dt = sorted(groups)[1]
            pti_info = {"s_per_iter": round(dt, 5), "s_per_iter_groups": [round(g, 5) for g in groups], "iters": 3 * n_it, "batch": 1, "resolution": 1024, "loss": "L2", "optimizer": "Adam (fused, capturable)",
                        "trainable_params": int(sum(p.numel() for p in params)), "loss_first": round(l0, 4), "loss_last": round(lN.item(), 4),
                        # forward + data gradient + weight gradient of every 3x3 modulated conv = 3 x the forward's algorithmic work (148.52 GFLOP
                        # per face), all of it in split-bf16 (3 bf16 MFMAs per product): the step's roofline is the MFMA one
                        "roofline": {"bound": "mfma", "achieved": round(3 * 148.52e9 / dt / 1e12, 2), "peak": round(BF16_MATRIX_PEAK_TFLOPS / 3.0, 1), "unit": "TFLOP/s",
                                     "frac": round(3 * 148.52e9 / dt / 1e12 / (BF16_MATRIX_PEAK_TFLOPS / 3.0), 4),
                                     "algorithmic_gflop_per_step": round(3 * 148.52, 1),
                                     "what": "3 x 148.52 algorithmic GFLOP per bs = 1 step over the replayed step's wall time, against 2500 / 3 TFLOP/s (split-bf16)"},
                        "how": "whole step (forward, backward, weight re-layout, Adam) as one hipGraph; synthesis gradients from csrc/modconv_bwd.hip "
                               "+ csrc/gemm_sb.hip (hand-written split-bf16 MFMA GEMM, implicit weight gradient; no library GEMM in the step) (BASELINE configs[3], one frame)"}
            # the loop of configs[3] itself on a short clip: passes over the frames, one optimiser step per frame, eroded maps, foreground-weighted
            # L2 (pti.tune_clip: training/video_swap_ft_coach.py:242-317); the first two steps run eagerly, the rest replays one captured step
            del step
            nf, passes = 32, max(1, args.pti_passes)      # BASELINE configs[3]'s clip length; --pti-passes 200 = its stated size
            vecs = torch.from_numpy(seeded.seeded_array(42, "vecs", (nf, 12, 1280), dist="normal")).to(dev)
            labs = torch.from_numpy(seeded.blocky_labels(43, nf, 12, 512, 16)).to(dev).to(torch.uint8)
            imgs = torch.tanh(torch.from_numpy(seeded.seeded_array(44, "imgs", (nf, 3, 1024, 1024), dist="normal"))).to(dev)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            hist = pti.tune_clip(tnet, topt, imgs, labs, vecs, steps=passes, erode_radius=3)
            torch.cuda.synchronize()
            pti_info["clip_loop"] = {"frames": nf, "passes": passes, "optimizer_steps": nf * passes, "seconds": round(time.perf_counter() - t2, 4),
                                     "mean_loss_per_pass": [round(h, 4) for h in (hist if len(hist) <= 10 else hist[:5] + hist[-5:])],
                                     "seconds_per_pass": round((time.perf_counter() - t2) / passes, 4),
                                     "what": "pti.tune_clip on configs[3]'s 32-frame clip at 1024 x 1024: erode_mask radius 3 + foreground-weighted L2, the first 2 steps eagerly on the "
                                             "capture stream, then one captured step replayed per frame (includes the capture); BASELINE configs[3] is 200 such passes, sharded over 4 GPUs "
                                             "with averaged gradients"}
            del topt, tnet, params, vecs, labs, imgs
            torch.cuda.empty_cache()
        except Exception as e:      # noqa: BLE001 - secondary measurement
            pti_info = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- BASELINE configs[3] on several GPUs (--gpus N > 1): the clip loop block-sharded over the ranks, one optimiser step per ROUND (every rank's i-th frame) on
    # gradients averaged over the ranks (pti.sync_gradients: the one collective of multi-GPU PTI, an all-reduce of ~111 MB of generator gradients per step over RCCL)
    if world > 1 and not args.no_pti:
        # Every collective of this section (barrier, all-reduce, tune_clip's gradient exchange) needs ALL ranks: a rank that failed while setting up (out of memory
        # on the clip + a trainable net, say) must not leave the others waiting in them.  So: set up and take one LOCAL optimiser step first (no collective: the
        # step's memory high-water mark is reached here), then agree on success (MIN over the ranks of an ok flag) and skip the section on every rank unless all are ok.
        setup_err = None
        try:
            from e4s2024_amd import pti
            torch.cuda.synchronize()
            tnet = Net3(_ap.Namespace(**{**vars(opts), "train_G": True}))
            seeded.apply_seeded(tnet, 4, "net3")
            tnet = tnet.to(dev).train()
            tnet.latent_avg = net.latent_avg
            params = pti.trainable_parameters(tnet)
            topt = torch.optim.Adam(params, lr=1e-3)
            nf, passes = 32, max(1, args.pti_passes)
            from e4s2024_amd.runner import shard_range as _sr
            lo_f, hi_f = _sr(nf, rank, world)
            vecs = torch.from_numpy(seeded.seeded_array(42, "vecs", (nf, 12, 1280), dist="normal")).to(dev)
            labs = torch.from_numpy(seeded.blocky_labels(43, nf, 12, 512, 16)).to(dev).to(torch.uint8)
            imgs = torch.tanh(torch.from_numpy(seeded.seeded_array(44, "imgs", (nf, 3, 1024, 1024), dist="normal"))).to(dev)
            pti.tune_clip(tnet, topt, imgs[:1], labs[:1], vecs[:1], steps=1, erode_radius=3, graphed=False,
                          local_only=True)           # one local step on frame 0 (the same on every rank: parameters stay identical), no collective
            torch.cuda.synchronize()
        except Exception as e:      # noqa: BLE001 - secondary measurement
            setup_err = f"{type(e).__name__}: {e}"[:300]
        ok = torch.tensor([0 if setup_err else 1], device=dev, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        all_ok = bool(ok.item())
        if not all_ok:
            pti_info = {"error": setup_err or "skipped: another rank failed to set the section up"}
    if world > 1 and not args.no_pti and all_ok:
        try:
            pti.tune_clip(tnet, topt, imgs, labs, vecs, steps=1, erode_radius=3)             # warm-up pass (weight caches, allocator, RCCL buffers)
            torch.cuda.synchronize()
            dist.barrier()
            t2 = time.perf_counter()
            hist = pti.tune_clip(tnet, topt, imgs, labs, vecs, steps=passes, erode_radius=3)
            torch.cuda.synchronize()
            dist.barrier()
            tt = torch.tensor([time.perf_counter() - t2], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            secs = float(tt.item())
            rounds = -(-nf // world)
            pti_info = {"ranks": world, "frames": nf, "passes": passes, "rounds_per_pass": rounds, "optimizer_steps": rounds * passes, "seconds": round(secs, 4),
                        "seconds_per_pass": round(secs / passes, 4), "s_per_iter": round(secs / (passes * rounds), 5), "batch": 1, "resolution": 1024, "loss": "L2 (foreground-weighted, eroded maps)",
                        "mean_loss_per_pass": [round(h, 4) for h in (hist if len(hist) <= 10 else hist[:5] + hist[-5:])],
                        "collective": "all-reduce (average) of the generator's gradients once per round (pti.sync_gradients over RCCL); one optimiser step per round on identical parameters",
                        "what": "pti.tune_clip on configs[3]'s 32-frame clip at 1024 x 1024, frames block-sharded over the ranks (rank r tunes on shard_range(32, r, N)); eager steps "
                                "(the graph-captured step has no gradient exchange); BASELINE configs[3] is 200 such passes on 4 GPUs — a batch-of-N step, not the reference's N "
                                "sequential steps: no parity claim for this mode (SURVEY section 8e)"}
            del topt, tnet, params, vecs, labs, imgs
            torch.cuda.empty_cache()
        except Exception as e:      # noqa: BLE001 - secondary measurement
            pti_info = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        faces = bs * world * args.steps
        value = faces / elapsed
        # ---- roofline of the dominant kernel
        # share of region-uniform 16 x 16 output blocks of the masked up layers under THESE masks (they run in the block kernel)
        ufrac = {}
        if ops.UP_BLOCKS and ops.MODCONV_MODE == "sb":
            labd = ops.mask_to_labels(mask)
            for res in (64, 128, 256):
                ub, _ = ops.uniform_blocks(labd, res, res, 12)
                # (share under one region, share made of four uniform 8 x 8 sub-blocks): 2.0x / 2.5x their algorithmic MACs
                ufrac[res] = (float((ub < 12).float().mean().item()), 0.0)
        fl, fl_exec = conv3x3_flops_per_face(want_executed=True, uniform_frac=ufrac)
        dom = max(ksum, key=lambda k: ksum[k][1]) if ksum else None
        roof = None
        if dom:
            calls, tot_ms = ksum[dom]
            per_launch_flops = fl[dom] * bs * args.steps / calls
            avg_ms = tot_ms / calls
            ach = per_launch_flops / (avg_ms * 1e-3) / 1e12
            all_ms = sum(v[1] for v in ksum.values())
            all_fl = sum(fl.values()) * bs * args.steps
            sb = ops.MODCONV_MODE == "sb"
            # split arithmetics: every algorithmic multiply-add costs `mfma_cost_per_product` bf16-MFMA multiply-adds of matrix-pipe time (3 for
            # split-bf16, 1.667 for the mx kernel's f16 + 2 x MX fp6), so the ceiling for ALGORITHMIC FLOPs is the dense bf16 MFMA peak divided by
            # it; exact mode is priced against the fp32 MFMA peak.
            cost = mfma_cost_per_product(dom)
            peak = BF16_MATRIX_PEAK_TFLOPS / cost if sb else FP32_MATRIX_PEAK_TFLOPS
            # the whole job against ITS ceiling: every kernel's algorithmic FLOPs at its own arithmetic's cost (the 0.4 GFLOP of 1x1 ToRGB convs: fp32)
            job_cost = sum(fl[k] * (mfma_cost_per_product(k) if sb else 16.0) for k in fl) / sum(fl.values())
            job_peak = BF16_MATRIX_PEAK_TFLOPS / job_cost
            job_ach = value * 148.52e9 / 1e12 / world
            roof = {"bound": "mfma", "kernel": dom,
                    **({"kernel_launches": "region_modconv_mx_kernel<1, ...> (csrc/modconv_mx.hip) and, for masked up layers whose launch fills the chip with 64-channel tiles "
                                           "(512->256 @64 at this batch), region_upconv_mx4_kernel (csrc/modconv_mx4.hip): the same tile code or its four-parity form, "
                                           "chosen per workgroup — profiler tables list the two names, this object counts them as one kernel"}
                       if (ops.UP_MX4 and dom and dom.startswith("region_modconv_mx_kernel<1")) else {}),
                    "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": _pmc_traffic(dom)[0], "traffic_source": _pmc_traffic(dom)[1],
                    "peak_basis": ((f"dense bf16 MFMA 2500 TFLOP/s / {cost:.3f} bf16-MFMA multiply-adds of matrix-pipe time per fp32-accurate product ("
                                    + ("f16 MFMA + 2 MX-fp6 MFMAs at 4x the rate, 24 of 32 K slots used" if cost < 3 else "split-bf16: 3 bf16 MFMAs") + ")") if sb else "fp32 MFMA 157.3 TFLOP/s"),
                    "frac_on_split_bf16_basis": round(ach / (BF16_MATRIX_PEAK_TFLOPS / 3.0), 4) if sb else None,
                    "whole_job_frac": round(job_ach / job_peak, 4), "whole_job_peak": round(job_peak, 1), "whole_job_achieved": round(job_ach, 2),
                    "whole_job_basis": f"148.52 algorithmic GFLOP per face x `value` against 2500 TFLOP/s / {job_cost:.3f} (FLOP-weighted cost of the arithmetic each 3x3 layer runs in)",
                    "whole_job_frac_on_split_bf16_basis": round(job_ach / (BF16_MATRIX_PEAK_TFLOPS / 3.0), 4) if sb else None,
                    "vs_fp32_mfma_peak": round(ach / FP32_MATRIX_PEAK_TFLOPS, 3),
                    "measured_over": ("the one-stream pass of the same K steps inside this run (`one_stream`): between its own HIP events a kernel shows its rate only "
                                      "while it has the chip to itself; `in_overlapped_region` = the same launches where two batches share the chip: "
                                      + ("an instrumented repeat of the K overlapped steps right behind the timed region (the timed steps carry no events: "
                                         "14 events per step cost `value` 1.0 % and its run-to-run spread, profiles/r05_ab_timed_events.txt)" if bare
                                         else "the timed region itself (--timed-events dominant)"))
                    if kt_overlap is not None else "the timed region (one stream)",
                    "in_overlapped_region": _overlap_view(kt_overlap, dom, per_launch_flops, peak),
                    "launches_per_step": calls // args.steps, "avg_launch_ms": round(avg_ms, 4),
                    "algorithmic_gflop_per_launch": round(per_launch_flops / 1e9, 3),
                    # context for `frac` (which prices ALGORITHMIC work against the nominal peak): what the kernel executes, and what this
                    # instruction mix sustains on this board with random operands
                    "executed_over_algorithmic": round(fl_exec[dom] / fl[dom], 3),
                    "executed_frac_of_nominal_peak": round(ach * fl_exec[dom] / fl[dom] / peak, 4) if sb else None,
                    "executed_frac_of_measured_sustained": round(ach * fl_exec[dom] / fl[dom] / (SUSTAINED_BF16_TFLOPS_RANDOM_DATA / cost), 4) if sb else None,
                    # the same ratio layer by layer (one launch per layer and step): the same-resolution layers run every algorithmic MAC once,
                    # the up-sampling layers execute the parity-composed form at 4x their algorithmic (transposed-conv) MACs
                    "by_layer": _by_layer(kt_for_layers, dom, bs, peak, ufrac),
                    "in_run_ab": in_run_ab,
                    "all_modconv3x3": {"achieved": round(all_fl / (all_ms * 1e-3) / 1e12, 2), "ms_per_step": round(all_ms / args.steps, 3),
                                       "by_kernel_ms_per_step": {k: round(v[1] / args.steps, 3) for k, v in sorted(ksum.items())}}}
        # ---- CPU baseline: the faithful 12-pass oracle on one face
        cpu = None
        if sd_cpu is not None:
            from oracle import e4s_oracle as O
            # 16 threads: measured best on the 256-thread EPYC 9575F GPU host (8: 5.0 s, 16: 3.4 s, 32: 3.7 s, 64: 5.7 s on a
            # 256x256 generator; torch's default of all 256 threads took 370 s for one 1024x1024 face) — tools/cpu_threads_probe.py
            torch.set_num_threads(min(16, os.cpu_count() or 1))
            c1, m1 = codes[:1].cpu(), mask[:1].cpu()
            runs = []
            for _ in range(3):                       # 3 x one face: ~15 s of CPU work (the first run also warms the thread pool)
                t1 = time.perf_counter()
                with torch.no_grad():
                    ref, _ = O.generator_forward(sd_cpu, c1, m1, None)
                runs.append(time.perf_counter() - t1)
            dt = sorted(runs)[1]
            err = (img[:1].cpu() - ref).abs().max().item()
            cpu = {"value": round(1.0 / dt, 4), "unit": "faces/s", "cores": torch.get_num_threads(), "host_hardware_threads": os.cpu_count(), "kind": "port",
                   "sample": f"3 x 1 face (bs=1) of the same workload through oracle.generator_forward (12 region passes per masked layer), "
                             f"median {dt:.1f} s (runs {', '.join(f'{r:.1f}' for r in runs)} s)",
                   "max_abs_pixel_diff_vs_gpu": float(f"{err:.3e}")}
        line = {
            "metric": "1024x1024 faces/sec (StyleGAN2 regional synthesis, gen_img)", "value": round(value, 3), "unit": "faces/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32" if ops.MODCONV_MODE != "sb" else "f16+mxfp6x2 (masked 3x3 layers >= 32^2) / bf16x3 (all other layers)" if ops.mx_arith() == 1 else "bf16x3"),
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: StyleGAN2 1024x1024 synthesis from random W+ (Net3.gen_img, randomize_noise=False), "
                                   f"12-region {args.labels} masks (a fresh one-hot mask tensor per step: the mask -> region-map conversion is timed), batch={bs}/GPU; arithmetic = "
                                   + (("split-bf16: fp32 operands split into bf16 hi+lo, 3 bf16 MFMAs per product, fp32 accumulate (max-abs pixel error "
                                       "6e-5 vs the reference; plain bf16 would miss the 1e-3 bar)"
                                       + ("; the masked 3x3 layers of width >= 32: f16 MFMA for a1*w1 plus two block-scaled MX-fp6 MFMAs for the cross terms "
                                          "(csrc/modconv_mx.hip; 2e-4 on the pixels)" if ops.mx_arith() == 1 else "")) if ops.MODCONV_MODE == "sb" else "exact fp32 MFMA"),
                       "batch_per_gpu": bs, "global_batch": bs * world, "resolution": 1024, "regions": 12, "parallelism": f"frames x{world}",
                       "streams_per_gpu": args.streams,
                       "step_overlap": (f"consecutive steps (independent batches) alternate over {args.streams} HIP streams: the latency-bound 4^2-32^2 layers of a "
                                        "batch run under the large layers of the one before; all K steps complete inside the timed region") if args.streams > 1 else "none"},
            "one_stream": one_stream, "soak": soak,
            "f16_range": {"overflowed_in_the_measured_passes": bool(f16_overflowed), "passes_rerun_in_split_bf16": int(ops.mx_fallbacks),
                          "what": "ops.MxGuard: the kernels of the f16-based arithmetic bump a device counter when a value leaves the f16 range; the benchmark brackets "
                                  "its synthesis passes with one guard (no per-pass host synchronisation) and swap_batch / the clip loop re-run a batch that moved it"},
            "roofline": roof, "cpu_baseline": cpu, "full_swap": full_swap, "pti": pti_info, "clip": clip_info, "mask_sensitivity": mask_sens,
            "algorithmic_gflop_per_face": 148.52,
            "job_algorithmic_tflops_per_gpu": round(value * 148.52e9 / 1e12 / world, 2),
        }
        # ---- output.  The driver keeps the LAST 2 000 characters of stdout and, of the parsed line, scalars only: everything it must see (both headline metrics of
        # BASELINE.json, PTI, clip, the stage split) is therefore a flat scalar of ONE compact final line; the bulky diagnostics (by_layer, in_run_ab, the `what` /
        # `how` texts) go to stderr and to gpurun_out/bench_detail.json.  stdout carries exactly one JSON line.
        detail_path = os.path.join(ROOT, "gpurun_out", "bench_detail.json")
        try:
            os.makedirs(os.path.dirname(detail_path), exist_ok=True)
            with open(detail_path, "w") as f:
                json.dump(line, f)
        except OSError:
            detail_path = None
        print("bench detail: " + json.dumps(line), file=sys.stderr, flush=True)
        print(json.dumps(_compact_line(line, one_stream, ksum, args.steps, detail_path)), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
