"""Error budget experiment (CPU), round 3: cheaper-than-three-bf16-MFMA arithmetic for the modulated 3x3 convolutions.

Same harness as emulate_split_bf16.py (the oracle's faithful twelve-pass path with the product a*b of every modulated 3x3
convolution replaced by an emulation of what the matrix pipe would compute, fp32 accumulation), but several schemes in one run,
all against ONE fp32 oracle evaluation, with the cost of each in bf16-MFMA equivalents per product (f16 = bf16 rate; MX-scaled
fp8 = 2x; MX-scaled fp6 / fp4 = 4x: MI355X_MICROARCH.md).  a = x*s (activation side, split per use), w = scale*W (weight
side, split once).

    bf16x3      a_hi w_hi + a_hi w_lo + a_lo w_hi, bf16 RNE                                      cost 3     (the shipped scheme)
    f16x2a      (a1 + a2) w1, f16 RNE                                                             cost 2     
    f16x2w      a1 (w1 + w2)                                                                      cost 2
    f16x1       a1 w1                                                                             cost 1
    f16x3       a1 w1 + a1 w2 + a2 w1                                                             cost 3
    f16+f8x2    a1 w1 in f16, the two cross terms a1 w2 and a2 w1 in e4m3 x e4m3 with ONE fixed
                power-of-two scale per operand kind (activations 2^3, their residuals 2^14,
                weights 2^10, their residuals 2^21: no block scales)                              cost 2
    f16+mxf8x2  the cross terms in MX fp8 e4m3 (a shared power-of-two scale per 32 channels)     cost 2
    f16+f6x2    the cross terms in MX fp6 e2m3 (a shared power-of-two scale per 32 channels)     cost 1.5
    f16+f4x2    the cross terms in MX fp4 e2m1                                                    cost 1.5
    f16+f6x2k   as f16+f6x2 with the kernel's cheap scale rule for the activation residual: the
                exponent of the block's largest |a|, 11 binades down (|a - f16(a)| <= 2^(E-11))      cost 1.5
    bf16+f8x2   bf16 main term, e4m3 cross terms                                                  cost 2

usage: python tests/experiments/emulate_split_variants.py [size] [labels: blocky|iid] [mode,mode,...]"""
import json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from e4s2024_amd import seeded
from oracle import e4s_oracle as O


def bf16(t): return t.to(torch.bfloat16).float()
def f16(t): return t.to(torch.float16).float()
def e4m3(t): return t.to(torch.float8_e4m3fn).float()


def _mx(t, cdim, mant_bits, emax, emin=0, scale_from=None, scale_shift=0):
    """Block-scaled (32 along the channel dim) round to a tiny float with `mant_bits` mantissa bits, largest exponent `emax` and
    smallest normal exponent `emin` (e2m3: 3, 2, 0 -> max 7.5; e2m1: 1, 2, 0 -> max 6; e4m3: 3, 8, -6 -> max 448 (480 here: the
    saturating top code is not modelled)): the OCP MX rule, scale = 2^(floor(log2(amax)) - emax), values RNE, saturating."""
    t = t.movedim(cdim, -1)
    shp = t.shape
    c = shp[-1]
    pad = (-c) % 32
    if pad:
        t = F.pad(t, (0, pad))
    g = t.reshape(*t.shape[:-1], -1, 32)
    if scale_from is None:
        amax = g.abs().amax(-1, keepdim=True).clamp_min(1e-38)
    else:       # the block scale is derived from ANOTHER tensor's block maxima (the kernel scales a - f16(a) by the exponent of max |a|, 11 binades down)
        u = scale_from.movedim(cdim, -1)
        if pad:
            u = F.pad(u, (0, pad))
        amax = u.reshape(*u.shape[:-1], -1, 32).abs().amax(-1, keepdim=True).clamp_min(1e-38)
    sc = torch.exp2(torch.floor(torch.log2(amax)) - emax + scale_shift)
    v = g / sc
    # RNE onto the grid: exponent of each value clamped to [0 (subnormal step), emax]
    e = torch.floor(torch.log2(v.abs().clamp_min(1e-38))).clamp(emin, emax)
    step = torch.exp2(e - mant_bits)
    q = torch.round(v / step) * step
    lim = (2 - 2.0 ** -mant_bits) * 2.0 ** emax
    q = q.clamp(-lim, lim) * sc
    q = q.reshape(*t.shape)[..., :c].reshape(shp)
    return q.movedim(-1, cdim)


def e2m3(t, cdim): return _mx(t, cdim, 3, 2)
def e2m1(t, cdim): return _mx(t, cdim, 1, 2)
def mxe4m3(t, cdim): return _mx(t, cdim, 3, 8, -6)


COST = {"bf16x3": 3, "f16x2a": 2, "f16x2w": 2, "f16x1": 1, "f16x3": 3, "f16+f8x2": 2, "f16+mxf8x2": 2, "f16+f6x2": 1.5, "f16+f6x2k": 1.5, "f16+f4x2": 1.5,
        "bf16+f8x2": 2, "fp32": 16}
MODE = "bf16x3"
RANGE = {"a_max": 0.0, "a_min_nz": 1e30, "w_max": 0.0}


def terms(a, w, a_cdim, w_cdim):
    """[(activation operand, weight operand), ...] whose convolutions are summed in fp32."""
    m = MODE
    if m == "fp32":
        return [(a, w)]
    if m == "bf16x3":
        ah, wh = bf16(a), bf16(w)
        al, wl = bf16(a - ah), bf16(w - wh)
        return [(ah, wh), (ah, wl), (al, wh)]
    if m == "bf16+f8x2":
        ah, wh = bf16(a), bf16(w)
        S = 2.0 ** 8
        al, wl = e4m3((a - ah) * S) / S, e4m3((w - wh) * S) / S
        return [(ah, wh), (e4m3(ah), wl), (al, e4m3(wh))]
    RANGE["a_max"] = max(RANGE["a_max"], float(a.abs().max()))
    RANGE["w_max"] = max(RANGE["w_max"], float(w.abs().max()))
    a1, w1 = f16(a), f16(w)
    if m == "f16x1":
        return [(a1, w1)]
    a2, w2 = a - a1, w - w1
    if m == "f16x2a":
        return [(a1, w1), (f16(a2), w1)]
    if m == "f16x2w":
        return [(a1, w1), (a1, f16(w2))]
    if m == "f16x3":
        return [(a1, w1), (a1, f16(w2)), (f16(a2), w1)]
    if m == "f16+f8x2":
        fx = lambda t, k: e4m3(t * 2.0 ** k) / 2.0 ** k
        return [(a1, w1), (fx(a1, 3), fx(w2, 21)), (fx(a2, 14), fx(w1, 10))]
    if m == "f16+mxf8x2":
        return [(a1, w1), (mxe4m3(a1, a_cdim), mxe4m3(w2, w_cdim)), (mxe4m3(a2, a_cdim), mxe4m3(w1, w_cdim))]
    if m == "f16+f6x2":
        return [(a1, w1), (e2m3(a1, a_cdim), e2m3(w2, w_cdim)), (e2m3(a2, a_cdim), e2m3(w1, w_cdim))]
    if m == "f16+f6x2k":
        return [(a1, w1), (e2m3(a1, a_cdim), e2m3(w2, w_cdim)), (_mx(a2, a_cdim, 3, 2, scale_from=a1, scale_shift=-11), e2m3(w1, w_cdim))]
    if m == "f16+f4x2":
        return [(a1, w1), (e2m1(a1, a_cdim), e2m1(w2, w_cdim)), (e2m1(a2, a_cdim), e2m1(w1, w_cdim))]
    raise ValueError(m)


_orig = O.modulated_conv2d


def patched(x, style, weight, mw, mb, demodulate=True, upsample=False, blur_kernel=None):
    if weight.shape[-1] != 3:
        return _orig(x, style, weight, mw, mb, demodulate, upsample, blur_kernel)
    bs, cin, h, w_ = x.shape
    _, cout, _, k, _ = weight.shape
    scale = 1.0 / math.sqrt(cin * k * k)
    s = O.equal_linear(style, mw, mb)
    B = x * s.view(bs, cin, 1, 1)
    if upsample:  # the mathematically equivalent un-fused form: split the transposed-conv operands, blur in fp32
        A = (scale * weight)[0].transpose(0, 1).contiguous()      # [cin, cout, k, k]
        out = sum(F.conv_transpose2d(b, a, stride=2) for b, a in terms(B, A, 1, 0))
        out = O.upfirdn2d(out, blur_kernel, pad=(1, 1))
    else:
        A = (scale * weight)[0]                                    # [cout, cin, k, k]
        out = sum(F.conv2d(b, a, padding=k // 2) for b, a in terms(B, A, 1, 1))
    if demodulate:
        d = torch.rsqrt(((scale * weight) * s.view(bs, 1, cin, 1, 1)).pow(2).sum([2, 3, 4]) + 1e-8)
        out = out * d.view(bs, cout, 1, 1)
    return out


def main():
    global MODE
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    labels = sys.argv[2] if len(sys.argv) > 2 else "blocky"
    modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["bf16x3", "f16x2a", "f16x2w", "f16x1", "f16+f8x2", "f16+f6x2"]
    man = json.load(open(os.path.join(os.path.dirname(__file__), "..", "golden", "manifest.json")))
    if size == 1024:
        tm = {k: v for k, v in man["net3_1024_rli13"].items() if k.startswith("G.")}
        seed, rli = 4, 13
    else:
        tm = man[f"generator_{size}_rli{13 if size == 256 else 5}"]
        seed, rli = 21, (13 if size == 256 else 5)
    tmpl = {k: torch.empty(tuple(s), dtype=getattr(torch, d), device="meta") for k, (s, d) in tm.items()}
    sd = seeded.seeded_state_dict(tmpl, seed, "net3")
    nl = int(math.log2(size)) * 2 - 2
    ncls = 12
    codes = seeded.seeded_codes(1, 1, ncls, nl, seeded.seeded_latent_avg(2, nl))
    lab = seeded.blocky_labels(3, 1, ncls, 512, 16) if labels == "blocky" else seeded.iid_labels(3, 1, ncls, 512)
    mask = seeded.labels_to_onehot(lab, ncls)
    torch.set_num_threads(8)
    out = {}
    with torch.no_grad():
        t = time.time(); ref, _ = O.generator_forward(sd, codes, mask, None, size=size, remaining_layer_idx=rli)
        print(f"size {size} labels {labels}: fp32 oracle {time.time() - t:.1f}s |ref|max {ref.abs().max():.3f} rms {ref.pow(2).mean().sqrt():.3f}", flush=True)
        O.modulated_conv2d = patched
        for m in modes:
            MODE = m
            t = time.time(); emu, _ = O.generator_forward(sd, codes, mask, None, size=size, remaining_layer_idx=rli)
            d = (emu - ref).abs()
            out[m] = {"cost": COST[m], "max_abs": float(d.max()), "mean_abs": float(d.mean()),
                      "p99.9": float(d.flatten().kthvalue(int(d.numel() * 0.999)).values)}
            print(f"  {m:10s} cost {COST[m]:>4}  max-abs {d.max():.3e}  mean-abs {d.mean():.3e}  p99.9 {out[m]['p99.9']:.3e}  ({time.time() - t:.0f}s)", flush=True)
    print(json.dumps({"size": size, "labels": labels, "f16_operand_range": RANGE, "schemes": out}))


if __name__ == "__main__":
    main()
