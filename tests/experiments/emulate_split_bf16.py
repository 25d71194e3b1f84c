"""Error budget experiment (CPU): what would the synthesis pixels look like if every modulated 3x3 conv computed
a*b as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with bf16 operands and fp32 accumulation (3 bf16 MFMAs instead of 1 fp32 MFMA)?

Emulates the split on the oracle's faithful path: A = scale*W (weights side, split once), B = x*s_c (activation side, split per
use), product terms through fp32 convs.  Prints max-abs pixel error vs the plain fp32 oracle at a given resolution."""
import json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from e4s2024_amd import seeded
from oracle import e4s_oracle as O

MODE = sys.argv[2] if len(sys.argv) > 2 else "trunc"


def split(t):
    if MODE == "rne":
        hi = t.to(torch.bfloat16).float()
    else:  # truncation of the low 16 bits
        hi = (t.view(torch.int32) & -65536).view(torch.float32)
    lo = (t - hi).to(torch.bfloat16).float()
    return hi, lo


def modconv_split(x, style, weight, mw, mb, demodulate, upsample, blur):
    bs, cin, h, w = x.shape
    _, cout, _, k, _ = weight.shape
    scale = 1.0 / math.sqrt(cin * k * k)
    s = O.equal_linear(style, mw, mb)                      # [bs, cin]
    A = (scale * weight)[0]                                 # [cout, cin, k, k]
    if upsample:                                            # what the kernel multiplies: blur-composed parity kernels
        return None
    A_hi, A_lo = split(A)
    B = x * s.view(bs, cin, 1, 1)
    B_hi, B_lo = split(B)
    out = F.conv2d(B_hi, A_hi, padding=k // 2) + F.conv2d(B_hi, A_lo, padding=k // 2) + F.conv2d(B_lo, A_hi, padding=k // 2)
    if demodulate:
        d = torch.rsqrt(((scale * weight) * s.view(bs, 1, cin, 1, 1)).pow(2).sum([2, 3, 4]) + 1e-8)
        out = out * d.view(bs, cout, 1, 1)
    return out


_orig = O.modulated_conv2d


def patched(x, style, weight, mw, mb, demodulate=True, upsample=False, blur_kernel=None):
    if weight.shape[-1] == 3 and not upsample:
        return modconv_split(x, style, weight, mw, mb, demodulate, upsample, blur_kernel)
    if weight.shape[-1] == 3 and upsample:
        # emulate on the mathematically equivalent un-fused form: split the transposed-conv operands, blur in fp32
        bs, cin, h, w = x.shape
        _, cout, _, k, _ = weight.shape
        scale = 1.0 / math.sqrt(cin * k * k)
        s = O.equal_linear(style, mw, mb)
        A = (scale * weight)[0].transpose(0, 1).contiguous()       # [cin, cout, k, k]
        A_hi, A_lo = split(A)
        B = x * s.view(bs, cin, 1, 1)
        B_hi, B_lo = split(B)
        ct = lambda b, a: F.conv_transpose2d(b, a, stride=2)
        out = ct(B_hi, A_hi) + ct(B_hi, A_lo) + ct(B_lo, A_hi)
        out = O.upfirdn2d(out, blur_kernel, pad=(1, 1))
        if demodulate:
            d = torch.rsqrt(((scale * weight) * s.view(bs, 1, cin, 1, 1)).pow(2).sum([2, 3, 4]) + 1e-8)
            out = out * d.view(bs, cout, 1, 1)
        return out
    return _orig(x, style, weight, mw, mb, demodulate, upsample, blur_kernel)


size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
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
mask = seeded.labels_to_onehot(seeded.blocky_labels(3, 1, ncls, 512, 16), ncls)
torch.set_num_threads(8)
with torch.no_grad():
    t = time.time(); ref, _ = O.generator_forward(sd, codes, mask, None, size=size, remaining_layer_idx=rli); t1 = time.time() - t
    O.modulated_conv2d = patched
    t = time.time(); emu, _ = O.generator_forward(sd, codes, mask, None, size=size, remaining_layer_idx=rli); t2 = time.time() - t
d = (emu - ref).abs()
print(f"size {size} mode {MODE}: |ref|max {ref.abs().max():.3f}  max-abs err {d.max():.3e}  mean-abs err {d.mean():.3e}  p99.9 {d.flatten().kthvalue(int(d.numel()*0.999)).values:.3e}  (fp32 {t1:.1f}s, emu {t2:.1f}s)")
