"""Row f1 (interim): gradients through the drop-in modules on the GPU — fused HIP forward, stock-PyTorch backward
(ops._TorchBackward + torch_ref.py) — against autograd through the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import install_dropin, record_parity, template_from_manifest
from e4s2024_amd import ops, seeded
from oracle import e4s_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731


@pytest.fixture(scope="module")
def sg2():
    install_dropin()
    from models.stylegan2 import model
    return model


def test_generator64_parameter_and_code_gradients(sg2, manifest):
    """One PTI-style backward through Generator(64, remaining_layer_idx=5): d(loss)/d(codes) and d(loss)/d(every parameter)
    equal the oracle's autograd results (the trainable set of a PTI step, training/video_swap_ft_coach.py:297-299)."""
    size, rli, ncls, bs = 64, 5, 5, 2
    lab = seeded.blocky_labels(22, bs, ncls, 64, cells=8)
    n_latent = int(np.log2(size)) * 2 - 2
    codes = seeded.seeded_codes(23, bs, ncls, n_latent, seeded.seeded_latent_avg(2, n_latent))
    sd = seeded.seeded_state_dict(template_from_manifest(manifest["generator_64_rli5"]), 21, "net3")
    mask = seeded.labels_to_onehot(lab, ncls)
    wgt = T(np.random.RandomState(5).standard_normal((bs, 3, size, size)).astype(np.float32))

    # oracle autograd on CPU
    sd_o = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "noises" not in k and "kernel" not in k else v) for k, v in sd.items()}
    codes_o = codes.clone().requires_grad_(True)
    img_o, _ = O.generator_forward(sd_o, codes_o, mask, None, size=size, remaining_layer_idx=rli, split_layer_idx=5)
    (img_o * wgt).sum().backward()

    gen = sg2.Generator(size, 512, 8, split_layer_idx=5, remaining_layer_idx=rli)
    gen.load_state_dict({k[2:]: v for k, v in sd.items()})
    gen = gen.to(DEV).train()
    codes_g = codes.clone().to(DEV).requires_grad_(True)
    img, _, _ = gen([codes_g], None, mask.to(DEV), input_is_latent=True, randomize_noise=False)
    assert (img.detach().cpu() - img_o.detach()).abs().max().item() <= 5e-4
    (img * wgt.to(DEV)).sum().backward()

    # Tolerances: this random-weight network is ill-conditioned for gradients (a 1e-5 change of an activation flips leaky-relu branches
    # further down): the fp32 oracle itself is 1e-3..8e-3 (2e-2 on a cancelling scalar sum) away from an fp64 evaluation of the same
    # graph.  The tight per-layer gradient checks are in tests/test_backward_cpu.py; this test pins the wiring of every parameter.
    def rel(a, b):
        return (a.cpu() - b).abs().max().item() / max(1e-6, b.abs().max().item())
    assert rel(codes_g.grad, codes_o.grad) <= 3e-2
    scalar_scale = max(v.grad.abs().max().item() for k, v in sd_o.items() if k.endswith("noise.weight") and getattr(v, "grad", None) is not None)
    checked = 0
    for name, p in gen.named_parameters():
        go = sd_o["G." + name].grad
        if go is None:          # not on the path (e.g. the 8-layer style MLP when input_is_latent=True)
            assert p.grad is None or p.grad.abs().max().item() == 0
            continue
        assert p.grad is not None, name
        if go.numel() == 1:     # noise weights: sum of +/- terms, compare on the scale of the largest of them
            assert (p.grad.cpu() - go).abs().item() <= 3e-2 * scalar_scale, (name, p.grad.item(), go.item())
        else:
            assert rel(p.grad, go) <= 3e-2, (name, rel(p.grad, go))
        checked += 1
    assert checked >= 40


def test_local_mlp_gradients():
    install_dropin()
    from models.networks import LocalMLP
    rs = np.random.RandomState(7)
    m = LocalMLP(dim_component=24, dim_style=16, num_w_layers=3)
    x = T(rs.standard_normal((4, 24)).astype(np.float32))
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    h = torch.nn.functional.leaky_relu(O.equal_linear(x, sd["mlp.0.weight"], sd["mlp.0.bias"]), 0.01)
    ref = O.equal_linear(h, sd["mlp.2.weight"], sd["mlp.2.bias"]).view(-1, 3, 16)
    g = T(rs.standard_normal(tuple(ref.shape)).astype(np.float32))
    (ref * g).sum().backward()
    m = m.to(DEV)
    out = m(x.to(DEV))
    assert (out.detach().cpu() - ref.detach()).abs().max().item() <= 1e-4
    (out * g.to(DEV)).sum().backward()
    for k, p in m.named_parameters():
        assert (p.grad.cpu() - sd[k].grad).abs().max().item() <= 1e-4 * max(1.0, sd[k].grad.abs().max().item()), k


def test_pti_step_eager_and_graph_agree():
    """pti.pti_step (eager) and pti.GraphedPTIStep (the same step captured as one hipGraph) walk the same loss trajectory on a small
    Net3 (64x64 generator), and the loss goes down."""
    import types
    install_dropin()
    from models.networks import Net3
    from e4s2024_amd import pti, ops
    opts = types.SimpleNamespace(fsencoder_type="psp", remaining_layer_idx=5, num_seg_cls=12, out_size=64, train_G=True,
                                 start_from_latent_avg=True, learn_in_w=False)
    vec = T(seeded.seeded_array(41, "vec", (1, 12, 1280), dist="normal")).to(DEV)
    lab = T(seeded.blocky_labels(3, 1, 12, 64, 8)).to(DEV)
    target = torch.tanh(T(seeded.seeded_array(5, "img", (1, 3, 64, 64), dist="normal"))).to(DEV)

    def make():
        torch.manual_seed(0)
        net = Net3(opts)
        seeded.apply_seeded(net, 4, "net3")
        net = net.to(DEV).train()
        net.latent_avg = seeded.seeded_latent_avg(2, 10).to(DEV)
        return net

    net_a = make()
    opt_a = torch.optim.Adam(pti.trainable_parameters(net_a), lr=1e-3)
    losses_a = []
    for _ in range(6):
        torch.manual_seed(1)          # same noise draw every step in both variants
        loss, _ = pti.pti_step(net_a, opt_a, vec, lab, target)
        losses_a.append(loss.item())
    assert losses_a[-1] < losses_a[0]

    net_b = make()
    opt_b = torch.optim.Adam(pti.trainable_parameters(net_b), lr=1e-3, capturable=True, fused=True)
    step = pti.GraphedPTIStep(net_b, opt_b, vec, lab, target, randomize_noise=False, warmup=2)
    net_c = make()
    opt_c = torch.optim.Adam(pti.trainable_parameters(net_c), lr=1e-3)
    losses_b, losses_c = [], []
    for _ in range(2):                # the two warm-up steps of the capture, eagerly, for the comparison run
        codes = net_c.cal_style_codes(vec)
        rec, _, _ = net_c.gen_img(None, codes, lab, randomize_noise=False)
        l = torch.nn.functional.mse_loss(rec, target)
        opt_c.zero_grad(); l.backward(); opt_c.step()
    for _ in range(4):
        lb, _ = step(vec, lab, target)
        losses_b.append(lb.item())
        codes = net_c.cal_style_codes(vec)
        rec, _, _ = net_c.gen_img(None, codes, lab, randomize_noise=False)
        l = torch.nn.functional.mse_loss(rec, target)
        opt_c.zero_grad(); l.backward(); opt_c.step()
        losses_c.append(l.item())
    assert np.allclose(losses_b, losses_c, rtol=2e-3), (losses_b, losses_c)


def test_style_vector_optimisation_steps_follow_the_oracle_with_a_second_target():
    """W-space optimisation as a CHECKED loop (optimization.py:321-349, 422-454): four gradient steps on the per-region style vectors of a 64 x 64 Net3 —
    L2 against the target plus an ``extra_loss`` term (the hook the reference's recolouring second target enters through, video_swap_ft_coach.py:283-284:
    here the mean colour of the reconstruction against a given colour) — beside the same four steps through autograd on the CPU oracle
    (cal_style_codes + the faithful twelve-pass generator_forward): the loss at every step and the tuned vectors must agree.  Plain SGD on both sides
    so that the comparison is well conditioned (Adam's first steps move every entry by +-lr whatever its gradient's size: the sign of a near-zero
    gradient entry would decide a 2 lr difference; the Adam loop itself is the property test below)."""
    import types
    install_dropin()
    from models.networks import Net3
    from e4s2024_amd import pti
    opts = types.SimpleNamespace(fsencoder_type="psp", remaining_layer_idx=5, num_seg_cls=12, out_size=64, train_G=False,
                                 start_from_latent_avg=True, learn_in_w=False)
    net = Net3(opts)
    seeded.apply_seeded(net, 4, "net3")
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    la = seeded.seeded_latent_avg(2, 10)
    net = net.to(DEV).eval()
    net.latent_avg = la.to(DEV)
    vec0 = T(seeded.seeded_array(41, "vec", (1, 12, 1280), dist="normal"))
    lab = seeded.blocky_labels(3, 1, 12, 64, 8)
    mask = seeded.labels_to_onehot(lab, 12)
    target = torch.tanh(T(seeded.seeded_array(5, "img", (1, 3, 64, 64), dist="normal")))
    colour = torch.tensor([0.3, -0.2, 0.1])

    def extra(dev):
        c = colour.to(dev).view(1, 3)
        return lambda recon, tgt: 0.5 * (recon.mean((2, 3)) - c).pow(2).mean()
    # device
    latent = vec0.clone().to(DEV).requires_grad_(True)
    opt = torch.optim.SGD([latent], lr=2.0)
    losses = [pti.style_vector_step(net, opt, latent, T(lab).to(DEV), target.to(DEV), extra_loss=extra(DEV), randomize_noise=False)[0].item()
              for _ in range(4)]
    # oracle
    lat_o = vec0.clone().requires_grad_(True)
    opt_o = torch.optim.SGD([lat_o], lr=2.0)
    losses_o = []
    for _ in range(4):
        opt_o.zero_grad()
        codes = O.cal_style_codes(sd, lat_o, la, 5)
        img, _ = O.generator_forward(sd, codes, mask, None, size=64, remaining_layer_idx=5)
        l = torch.nn.functional.mse_loss(img, target) + extra("cpu")(img, target)
        l.backward()
        opt_o.step()
        losses_o.append(l.item())
    rel = max(abs(a - b) / abs(b) for a, b in zip(losses, losses_o))
    dv = (latent.detach().cpu() - lat_o.detach()).abs().max().item()
    moved = (lat_o.detach() - vec0).abs().max().item()
    record_parity("w_optimisation.4_sgd_steps.loss_rel_vs_oracle", rel, 2e-3)
    record_parity("w_optimisation.4_sgd_steps.style_vectors_vs_oracle", dv / max(moved, 1e-12), 3e-2, note=f"relative to the largest move of an entry ({moved:.2e})")
    assert rel <= 2e-3 and dv <= 3e-2 * moved and moved > 0, (losses, losses_o, dv, moved)


def test_style_vector_optimisation_reduces_the_loss():
    """W-optimisation (optimization.py:321-349): gradient steps on the style vectors alone, network frozen."""
    import types
    install_dropin()
    from models.networks import Net3
    from e4s2024_amd import pti
    opts = types.SimpleNamespace(fsencoder_type="psp", remaining_layer_idx=5, num_seg_cls=12, out_size=64, train_G=False,
                                 start_from_latent_avg=True, learn_in_w=False)
    net = Net3(opts)
    seeded.apply_seeded(net, 4, "net3")
    net = net.to(DEV).eval()
    net.latent_avg = seeded.seeded_latent_avg(2, 10).to(DEV)
    before = {k: v.detach().clone() for k, v in net.state_dict().items()}
    latent = T(seeded.seeded_array(41, "vec", (1, 12, 1280), dist="normal")).to(DEV).requires_grad_(True)
    lab = T(seeded.blocky_labels(3, 1, 12, 64, 8)).to(DEV)
    target = torch.tanh(T(seeded.seeded_array(5, "img", (1, 3, 64, 64), dist="normal"))).to(DEV)
    opt = torch.optim.Adam([latent], lr=0.05)
    losses = [pti.style_vector_step(net, opt, latent, lab, target, randomize_noise=False)[0].item() for _ in range(8)]
    assert losses[-1] < 0.9 * losses[0], losses
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k]), k          # the network did not move


@pytest.mark.parametrize("ks,upsample,demod,shape", [(3, False, True, (2, 20, 24, 80, 72)), (3, True, True, (2, 12, 20, 18, 22)),
                                                     (1, False, False, (2, 24, 3, 33, 40)), (3, False, True, (1, 8, 8, 4, 4)),
                                                     (3, True, True, (1, 8, 16, 4, 4))])
@pytest.mark.parametrize("masked", [True, False])
def test_native_masked_conv_gradients_against_stock_pytorch_and_fp64(ks, upsample, demod, shape, masked):
    """csrc/modconv_bwd.hip (unfold / scale / fold kernels + fp32 GEMMs) against (a) the same layer written with stock PyTorch ops on the
    GPU and (b) the reference's own form — Σ_c modconv(x, style_c) ⊙ mask_c, then noise, bias, leaky-relu (model.py:389-423) — in fp64 on
    the CPU: gradients w.r.t. the input, the conv weight, the styles, the modulation weights, the noise weight and the activation bias.
    ``masked=False`` is a single-region layer (no label map: ``_SingleStyledConvGrad``).  Two native routes: the re-evaluating core (``_MaskedConvCore``, what ToRGB uses) and the one that differentiates from the saved
    forward value without re-evaluating the layer (``_MaskedStyledConvGrad``, what StyledConv uses).  Ragged sizes, more pixels than one
    fold chunk, and pixels that belong to no region (label 255)."""
    import torch.nn.functional as F
    from e4s2024_amd import ops, torch_ref
    bs, cin, cout, h, w = shape
    if not masked and cout < 16 and not ops.ALLOW_LIBRARY_BWD:
        # single-region layers this small have no native data-gradient kernel: by default their backward RAISES (no vendor library unasked)
        # and E4S_ALLOW_MIOPEN_BWD=1 admits aten.convolution_backward — both behaviours are part of the contract
        sc = ops._SingleStyledConvGrad
        xs = torch.randn(bs, cin, h, w, device=DEV, requires_grad=True)
        wm = torch.randn(bs, cout, cin, ks, ks, device=DEV, requires_grad=True)
        hh, ww = (2 * h, 2 * w) if upsample else (h, w)
        bl = None
        if upsample:
            bl = torch.tensor([1., 3., 3., 1.], device=DEV)
            bl = (bl[:, None] * bl[None, :]) / bl.sum() ** 2 * 4
        o = sc.apply(xs, wm, None, None, None, False, bl, torch.randn(bs, cout, hh, ww, device=DEV))
        with pytest.raises(NotImplementedError, match="E4S_ALLOW_MIOPEN_BWD"):
            o.sum().backward()
        ops.ALLOW_LIBRARY_BWD = True
        try:
            return test_native_masked_conv_gradients_against_stock_pytorch_and_fp64(ks, upsample, demod, shape, masked)
        finally:
            ops.ALLOW_LIBRARY_BWD = False
    nreg, sdim = (5 if masked else 1), 16
    g = torch.Generator().manual_seed(100 + ks + 2 * upsample + h)
    x = torch.randn(bs, cin, h, w, generator=g)
    weight = torch.randn(1, cout, cin, ks, ks, generator=g)
    styles = torch.randn(bs, nreg, sdim, generator=g)
    mod_w, mod_b = torch.randn(cin, sdim, generator=g), torch.ones(cin)
    nw, act_bias = torch.tensor([0.7]), torch.randn(cout, generator=g)
    ho, wo = (2 * h, 2 * w) if upsample else (h, w)
    noise = torch.randn(1, 1, ho, wo, generator=g)
    lab = torch.from_numpy(seeded.blocky_labels(7 + h, bs, nreg, max(ho, wo), cells=4))[:, :ho, :wo].contiguous().to(torch.uint8)
    lab[:, 0, :3] = 255                                   # no region
    if not masked:
        lab = None
    gout = torch.randn(bs, cout, ho, wo, generator=g)
    blur = torch.tensor([1., 3., 3., 1.])
    blur = (blur[:, None] * blur[None, :]) / blur.sum() ** 2 * 4
    mod_scale = 1.0 / np.sqrt(sdim)
    names = ("dx", "dweight", "dstyles", "dmod_w", "dnoise_weight", "dact_bias")

    def run(route, dev, dtype, fwd_out=None, tables=None):
        leaves = [t.to(dev, dtype).requires_grad_(True) for t in (x, weight, styles, mod_w, nw, act_bias)]
        old = ops.NATIVE_BWD
        ops.NATIVE_BWD = route != "stock"
        try:
            out = torch_ref.styled_conv(leaves[0], leaves[2], leaves[1], leaves[3], mod_b.to(dev, dtype), leaves[4], leaves[5], labels=None if lab is None else lab.to(dev),
                                        noise=noise.to(dev, dtype), act=True, upsample=upsample, blur=blur.to(dev, dtype) if upsample else None,
                                        demodulate=demod, mod_scale=mod_scale, mod_lr=1.0, fwd_out=fwd_out, tables=tables)
            grads = torch.autograd.grad(out, leaves, gout.to(dev, dtype))
        finally:
            ops.NATIVE_BWD = old
        return out.detach(), [t.cpu().double() for t in grads]

    def reference_fp64():                                  # the reference's twelve-pass form with its own per-region ModulatedConv2d
        leaves = [t.double().requires_grad_(True) for t in (x, weight, styles, mod_w, nw, act_bias)]
        xx, ww, st, mw, nwl, bl = leaves
        out = 0
        for c in range(nreg):
            s = F.linear(st[:, c], mw * mod_scale, mod_b.double())
            y = torch_ref._modulated(xx, s, ww, demod, upsample, blur.double() if upsample else None)
            out = out + (y * (lab == c)[:, None].double() if masked else y)
        out = F.leaky_relu(out + nwl * noise.double() + bl.view(1, -1, 1, 1), 0.2) * np.sqrt(2.0)
        return out.detach(), list(torch.autograd.grad(out, leaves, gout.double()))

    out_stock, stock = run("stock", DEV, torch.float32)
    out_core, core = run("core", DEV, torch.float32)
    out_saved, saved = run("saved", DEV, torch.float32, fwd_out=out_stock)
    # the same route with the style tables the forward kernels compute (e4s_style_demod) handed over: gradient = e4s_style_tables_bwd
    wsq = (weight[0] / np.sqrt(cin * ks * ks)).pow(2).sum((2, 3)).t().to(DEV).contiguous()          # [cin, cout], the layout of e4s_modconv_prep_weights
    s_hip, d_hip = ops.style_demod(styles.to(DEV), mod_w.to(DEV), mod_b.to(DEV), wsq if demod else None, cout)
    _, saved_tables = run("saved", DEV, torch.float32, fwd_out=out_stock, tables=(s_hip, d_hip, wsq))
    out_ref, ref = reference_fp64()
    assert torch.equal(out_saved, out_stock)                # the saved-forward route returns the value it was given
    oscale = out_ref.abs().max().item()
    assert (out_core.cpu().double() - out_ref).abs().max().item() / oscale <= 2e-5
    if masked:      # no region: only noise + bias pass through the activation
        passthrough = F.leaky_relu(act_bias.view(-1, 1) + 0.7 * noise[0, 0, 0, :3].view(1, -1), 0.2) * 2 ** 0.5
        assert (out_core[:, :, 0, :3].cpu() - passthrough[None]).abs().max().item() <= 1e-6
    for name, a, r in zip(names, saved_tables, ref):
        assert (a - r).abs().max().item() / max(1e-6, r.abs().max().item()) <= 3e-5, f"{name}: saved forward + saved tables vs fp64 reference form"
    for name, a, b, c, r in zip(names, core, saved, stock, ref):
        scale = max(1e-6, r.abs().max().item())
        assert (a - r).abs().max().item() / scale <= 3e-5, f"{name}: re-evaluating native core vs fp64 reference form"
        assert (b - r).abs().max().item() / scale <= 3e-5, f"{name}: saved-forward native route vs fp64 reference form"
        assert (c - r).abs().max().item() / scale <= 3e-5, f"{name}: stock PyTorch vs fp64 reference form"
    assert core[0].abs().max().item() > 0 and saved[4].abs().max().item() > 0


@pytest.mark.parametrize("masked", [True, False])
@pytest.mark.parametrize("with_skip", [True, False])
def test_native_torgb_gradients_against_stock_pytorch_and_fp64(masked, with_skip):
    """``ops._ToRGBGrad`` (gradients of ToRGB from the kernels of csrc/modconv_bwd.hip + the FIR kernel, no re-evaluation) against the
    stock-PyTorch form on the GPU and the reference's per-region form in fp64 (model.py:439-479): x, conv weight, styles, modulation
    weight, bias and the skip image."""
    import torch.nn.functional as F
    from e4s2024_amd import ops, torch_ref
    bs, cin, h, w, sdim = 2, 24, 36, 28, 16
    nreg = 5 if masked else 1
    g = torch.Generator().manual_seed(300 + masked + 2 * with_skip)
    x = torch.randn(bs, cin, h, w, generator=g)
    weight = torch.randn(1, 3, cin, 1, 1, generator=g)
    styles = torch.randn(bs, nreg, sdim, generator=g)
    mod_w, mod_b = torch.randn(cin, sdim, generator=g), torch.ones(cin)
    bias = torch.randn(1, 3, 1, 1, generator=g)
    skip = torch.randn(bs, 3, h // 2, w // 2, generator=g) if with_skip else None
    lab = torch.from_numpy(seeded.blocky_labels(9, bs, nreg, max(h, w), cells=4))[:, :h, :w].contiguous().to(torch.uint8) if masked else None
    if masked:
        lab[:, 1, 2:5] = 255
    gout = torch.randn(bs, 3, h, w, generator=g)
    k1 = torch.tensor([1., 3., 3., 1.])
    up_kernel = (k1[:, None] * k1[None, :]) / k1.sum() ** 2 * 4
    mod_scale = 1.0 / np.sqrt(sdim)

    def run(route, dev, dtype, fwd_out=None):
        leaves = [t.to(dev, dtype).requires_grad_(True) for t in (x, weight, styles, mod_w, bias) + ((skip,) if with_skip else ())]
        old = ops.NATIVE_BWD
        ops.NATIVE_BWD = route != "stock"
        try:
            out = torch_ref.to_rgb(leaves[0], leaves[2], leaves[5] if with_skip else None, leaves[1], leaves[3], mod_b.to(dev, dtype), leaves[4],
                                   labels=None if lab is None else lab.to(dev), up_kernel=up_kernel.to(dev, dtype), mod_scale=mod_scale, mod_lr=1.0,
                                   fwd_out=fwd_out)
            grads = torch.autograd.grad(out, leaves, gout.to(dev, dtype))
        finally:
            ops.NATIVE_BWD = old
        return out.detach(), [t.cpu().double() for t in grads]

    def reference_fp64():
        leaves = [t.double().requires_grad_(True) for t in (x, weight, styles, mod_w, bias) + ((skip,) if with_skip else ())]
        out = 0
        for c in range(nreg):
            s = F.linear(leaves[2][:, c], leaves[3] * mod_scale, mod_b.double())
            y = torch_ref._modulated(leaves[0], s, leaves[1], False, False, None)
            out = out + (y * (lab == c)[:, None].double() if masked else y)
        out = out + leaves[4]
        if with_skip:
            out = out + torch_ref.fir_resample(leaves[5], up_kernel.double(), up=2, pad=(2, 1))
        return out.detach(), list(torch.autograd.grad(out, leaves, gout.double()))

    out_stock, stock = run("stock", DEV, torch.float32)
    out_saved, saved = run("saved", DEV, torch.float32, fwd_out=out_stock)
    out_ref, ref = reference_fp64()
    assert torch.equal(out_saved, out_stock)
    assert (out_stock.cpu().double() - out_ref).abs().max().item() / out_ref.abs().max().item() <= 2e-5
    for name, a, b, r in zip(("dx", "dweight", "dstyles", "dmod_w", "dbias", "dskip"), saved, stock, ref):
        scale = max(1e-6, r.abs().max().item())
        assert (a - r).abs().max().item() / scale <= 3e-5, f"{name}: native ToRGB gradient vs fp64 reference form"
        assert (b - r).abs().max().item() / scale <= 3e-5, f"{name}: stock PyTorch vs fp64 reference form"


@pytest.mark.parametrize("update", ["fused_adam", "data_write"])
def test_weights_updated_behind_the_version_counter_are_seen_by_the_next_forward(sg2, manifest, update):
    """``torch.optim.Adam(fused=True)`` and ``p.data`` writes change a parameter without bumping its version counter (measured on this
    build: tools/probes/version_probe.py).  The re-laid-out weight copies of the kernels must not survive such an update: after a training
    forward + update, a ``no_grad`` forward equals the forward of a freshly built generator holding the updated weights.  Raw writes
    with no training forward in between are covered by ``e4s2024_amd.invalidate_weight_caches``."""
    size, rli, ncls, bs = 64, 5, 5, 1
    lab = seeded.blocky_labels(22, bs, ncls, 64, cells=8)
    n_latent = int(np.log2(size)) * 2 - 2
    codes = seeded.seeded_codes(23, bs, ncls, n_latent, seeded.seeded_latent_avg(2, n_latent)).to(DEV)
    sd = seeded.seeded_state_dict(template_from_manifest(manifest["generator_64_rli5"]), 21, "net3")
    mask = seeded.labels_to_onehot(lab, ncls).to(DEV)

    def build(state):
        g = sg2.Generator(size, 512, 8, split_layer_idx=5, remaining_layer_idx=rli)
        g.load_state_dict(state)
        return g.to(DEV)

    gen = build({k[2:]: v for k, v in sd.items()}).train()
    with torch.no_grad():
        before = gen([codes], None, mask, input_is_latent=True, randomize_noise=False)[0].clone()     # fills every weight cache
    params = [p for p in gen.parameters() if p.requires_grad]
    versions = [p._version for p in params]
    if update == "fused_adam":
        opt = torch.optim.Adam(params, lr=5e-2, fused=True)
        img = gen([codes], None, mask, input_is_latent=True, randomize_noise=False)[0]
        opt.zero_grad()
        img.square().mean().backward()
        opt.step()
    else:
        import e4s2024_amd
        with torch.no_grad():
            for p in params:
                p.data.mul_(1.05)
        assert e4s2024_amd.invalidate_weight_caches(gen) >= 10      # raw writes between two no_grad forwards leave no trace: the caller says so
    assert [p._version for p in params] == versions, "this build now bumps versions here: the test no longer exercises the hazard"
    with torch.no_grad():
        after = gen([codes], None, mask, input_is_latent=True, randomize_noise=False)[0]
        fresh = build(gen.state_dict())([codes], None, mask, input_is_latent=True, randomize_noise=False)[0]
    assert (after - before).abs().max().item() > 1e-3            # the update did change the image
    assert torch.equal(after, fresh)


def test_trained_weights_are_prepared_once_per_forward_and_never_reused_across_forwards(sg2):
    """``ops.one_forward``: inside one pass several call sites share one re-laid-out copy of a trained weight; the next pass (the weights may
    have been written behind the version counter in between) builds a new one; outside a pass nothing is shared."""
    from e4s2024_amd import ops
    conv = sg2.ModulatedConv2d(32, 32, 3, 512).to(DEV)
    assert conv.weight.requires_grad
    with ops.one_forward():
        a = conv._weights(True)
        b = conv._weights(True)
        assert a[0][0] is b[0][0] and a[1] is b[1]
    with torch.no_grad():
        conv.weight.data.mul_(2.0)                      # no version bump
    with ops.one_forward():
        c = conv._weights(True)
    assert c[0][0] is not a[0][0]
    assert torch.allclose(c[1], a[1] * 4.0, rtol=1e-5)  # wsq of the doubled weight
    d, e = conv._weights(True), conv._weights(True)     # outside a pass: rebuilt on every ask
    assert d[0][0] is not e[0][0]
    with torch.no_grad():                               # inference: ordinary version-keyed caching
        f, g = conv._weights(True), conv._weights(True)
    assert f[0][0] is g[0][0]


def test_pti_step_gradients_1024_vs_oracle_autograd(net3_sd):
    """BASELINE configs[3] size: the backward of ONE PTI step at 1024 x 1024 (cal_style_codes -> gen_img -> L2 loss -> backward,
    training/video_swap_ft_coach.py:268-299) on the native gradient kernels, against autograd through the CPU oracle
    (cal_style_codes + the faithful twelve-pass generator_forward) for the style vectors and a sampled subset of the trainable
    parameters that covers every kind on the path: masked / single-region conv weights at every resolution class, up-sampling and
    plain layers, modulation weights and biases, noise weights, activation biases, masked ToRGB and per-region MLPs (the single-region
    layers past remaining_layer_idx are frozen by the reference's constructor: the gradient only passes through them)."""
    from conftest import default_opts, record_parity
    install_dropin()
    from models.networks import Net3
    import torch.nn.functional as F
    la = seeded.seeded_latent_avg(2, 18)
    vec = T(seeded.seeded_array(41, "vec", (1, 12, 1280), dist="normal"))
    lab = seeded.blocky_labels(3, 1, 12, 512, 16)
    mask = seeded.labels_to_onehot(lab, 12)
    target = torch.tanh(T(seeded.seeded_array(5, "img", (1, 3, 1024, 1024), dist="normal")))
    subset = ["G.conv1.conv.weight", "G.conv1.conv.modulation.weight", "G.conv1.noise.weight", "G.conv1.activate.bias", "G.to_rgb1.conv.weight",
              "G.convs.0.conv.weight", "G.convs.3.conv.modulation.bias", "G.convs.5.conv.weight", "G.convs.6.conv.weight", "G.convs.7.conv.weight",
              "G.convs.7.noise.weight", "G.convs.8.conv.weight", "G.convs.8.conv.modulation.weight", "G.convs.9.conv.weight", "G.convs.9.activate.bias",
              "G.convs.10.conv.weight", "G.convs.10.noise.weight", "G.convs.11.conv.weight", "G.convs.11.conv.modulation.weight",
              "G.convs.11.activate.bias", "G.to_rgbs.2.conv.weight", "G.to_rgbs.3.bias", "G.to_rgbs.4.conv.weight", "G.to_rgbs.4.conv.modulation.weight",
              "MLPs.6.mlp.0.weight", "MLPs.6.mlp.2.weight", "MLPs.6.mlp.2.bias", "MLPs.1.mlp.0.bias"]
    # the reference's Net3(train_G=True) freezes everything past remaining_layer_idx (the single-region layers convs[12..15], to_rgbs[5..7]:
    # manifest "net3_1024_rli13_trainG_requires_grad_false"); the gradient still has to pass THROUGH them to reach every layer above
    frozen = ["G.convs.12.conv.weight", "G.convs.13.conv.weight", "G.convs.14.conv.weight", "G.convs.15.conv.weight", "G.to_rgbs.5.conv.weight",
              "G.to_rgbs.7.conv.weight"]
    assert all(k in net3_sd for k in subset)

    # ---- device: one PTI step's loss and backward on the drop-in Net3 (train_G=True), fixed noise buffers
    net = Net3(default_opts(train_G=True))
    net.load_state_dict(net3_sd)
    net = net.to(DEV).train()
    net.latent_avg = la.to(DEV)
    vec_g = vec.clone().to(DEV).requires_grad_(True)
    codes = net.cal_style_codes(vec_g)
    rec, _, _ = net.gen_img(None, codes, T(lab).to(DEV).to(torch.uint8), randomize_noise=False)
    loss = F.mse_loss(rec, target.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    named = dict(net.named_parameters())
    assert all(not named[k].requires_grad and named[k].grad is None for k in frozen)

    # ---- oracle autograd on the host
    sd_o = {k: (v.clone().requires_grad_(True) if k in subset else v) for k, v in net3_sd.items()}
    vec_o = vec.clone().requires_grad_(True)
    with torch.enable_grad():
        codes_o = O.cal_style_codes(sd_o, vec_o, la, 13)
        img_o, _ = O.generator_forward(sd_o, codes_o, mask, None)
        loss_o = F.mse_loss(img_o, target)
        loss_o.backward()
    dimg = (rec.detach().cpu() - img_o.detach()).abs().max().item()
    record_parity("pti1024.forward_pixels_vs_oracle", dimg, 1e-3)
    assert dimg <= 1e-3
    assert abs(loss.item() - loss_o.item()) <= 1e-4 * abs(loss_o.item())

    def rel(a, b):
        return (a.detach().cpu() - b).abs().max().item() / max(1e-12, b.abs().max().item())
    # Tolerances per parameter, from the measured worst of each (profiles/r04_final_parity.json; round 3 had a blanket 3e-2): 27 of the 28 sampled parameters
    # and the style vectors agree with autograd through the oracle to <= 1.1e-3 of their gradient's largest entry — bar 3e-3.  The exception is the weight of
    # the 4 -> 8 up layer (G.convs.0): its gradient sums over the 64 pixels of an 8 x 8 map, so ONE leaky-ReLU branch that flips under the forward's 1e-5-level
    # difference moves 1/64 of an entry (measured 9.6e-3; the fp32 oracle itself is ~1e-2 from an fp64 evaluation there) — bar 2e-2 for that layer only.
    # The tight checks of the kernels' arithmetic (3e-5 against fp64) are the per-layer tests above; this one pins wiring and scaling of every parameter kind.
    def tol_of(name):
        return 2e-2 if name.startswith("G.convs.0.") else 3e-3
    r = rel(vec_g.grad, vec_o.grad)
    record_parity("pti1024.grad.style_vectors.rel_vs_oracle", r, 3e-3)
    worst = ("style_vectors", r, r / 3e-3)
    assert r <= 3e-3
    scalar_scale = max(sd_o[k].grad.abs().max().item() for k in subset if k.endswith("noise.weight"))
    for k in subset:
        go, gg = sd_o[k].grad, named[k].grad
        assert go is not None and gg is not None, k
        if go.numel() == 1:
            r = (gg.cpu() - go).abs().item() / scalar_scale
        else:
            r = rel(gg, go)
        record_parity(f"pti1024.grad.{k}.rel_vs_oracle", r, tol_of(k))
        if r / tol_of(k) > worst[2]:
            worst = (k, r, r / tol_of(k))
        assert r <= tol_of(k), (k, r)
    record_parity("pti1024.grad.worst_share_of_its_bar", worst[2], 1.0, worst[0])


def test_pti_step_1024_runs_no_library_gemm_or_convolution(net3_sd):
    """one eager PTI step at 1024 x 1024 (BASELINE configs[3]'s unit) under torch.profiler — no rocBLAS / Tensile (``Cijk_``),
    no MIOpen kernel and no ``aten::mm / bmm / addmm / convolution*`` op anywhere in forward, backward or optimiser step."""
    from conftest import default_opts
    from torch.profiler import profile, ProfilerActivity
    from e4s2024_amd import pti
    install_dropin()
    from models.networks import Net3
    assert ops.NATIVE_BWD and not ops.ALLOW_LIBRARY_BWD
    net = Net3(default_opts(train_G=True))
    net.load_state_dict(net3_sd)
    net = net.to(DEV).train()
    net.latent_avg = seeded.seeded_latent_avg(2, 18).to(DEV)
    opt = torch.optim.Adam(pti.trainable_parameters(net), lr=1e-3, fused=True)
    vec = T(seeded.seeded_array(41, "vec", (1, 12, 1280), dist="normal")).to(DEV)
    lab = T(seeded.blocky_labels(3, 1, 12, 512, 16)).to(DEV).to(torch.uint8)
    target = torch.tanh(T(seeded.seeded_array(5, "img", (1, 3, 1024, 1024), dist="normal"))).to(DEV)
    pti.pti_step(net, opt, vec, lab, target)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        pti.pti_step(net, opt, vec, lab, target)
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    bad_ops = {"aten::mm", "aten::bmm", "aten::addmm", "aten::baddbmm", "aten::matmul", "aten::convolution", "aten::_convolution",
               "aten::convolution_backward", "aten::miopen_convolution", "aten::conv2d", "aten::conv_transpose2d", "aten::linear"}
    hit = [n for n in names if n in bad_ops or n.startswith("Cijk_") or "miopen" in n.lower() or "MIOpen" in n or "igemm" in n.lower()]
    assert not hit, hit
    assert any(n.startswith("e4s::") or "gemm_sb" in n or "mconv" in n for n in names), "the profile should show this library's kernels"


def test_graphed_pti_replay_then_eval_uses_fresh_weights():
    """a hipGraph replay of the optimiser step moves the parameters with no Python forward and no version bump.  The order
    replay -> no_grad preview -> replay -> no_grad preview must render each preview with the weights of that moment: compared with a
    freshly built generator loaded from the tuned state_dict (whose caches cannot be stale)."""
    import types
    install_dropin()
    from models.networks import Net3
    from e4s2024_amd import pti
    opts = types.SimpleNamespace(fsencoder_type="psp", remaining_layer_idx=5, num_seg_cls=12, out_size=64, train_G=True,
                                 start_from_latent_avg=True, learn_in_w=False)
    vec = T(seeded.seeded_array(41, "vec", (1, 12, 1280), dist="normal")).to(DEV)
    lab = T(seeded.blocky_labels(3, 1, 12, 64, 8)).to(DEV)
    target = torch.tanh(T(seeded.seeded_array(5, "img", (1, 3, 64, 64), dist="normal"))).to(DEV)

    def make(sd=None):
        net = Net3(opts)
        seeded.apply_seeded(net, 4, "net3")
        if sd is not None:
            net.load_state_dict(sd)
        net = net.to(DEV)
        net.latent_avg = seeded.seeded_latent_avg(2, 10).to(DEV)
        return net

    def preview(net):
        with torch.no_grad():
            return net.gen_img(None, net.cal_style_codes(vec), lab, randomize_noise=False)[0].clone()

    net = make().train()
    opt = torch.optim.Adam(pti.trainable_parameters(net), lr=5e-3, capturable=True, fused=True)
    step = pti.GraphedPTIStep(net, opt, vec, lab, target, randomize_noise=False, warmup=2)
    previews = []
    for _ in range(2):
        step(vec, lab, target)
        step(vec, lab, target)
        torch.cuda.synchronize()
        got = preview(net)                                   # builds (and caches) re-laid-out weights under no_grad
        fresh = make({k: v.detach().clone() for k, v in net.state_dict().items()}).eval()
        want = preview(fresh)
        assert torch.equal(got, want), "no_grad forward after a graph replay rendered with stale weight copies"
        previews.append(got)
    assert (previews[0] - previews[1]).abs().max().item() > 1e-4      # the weights really moved between the two previews


def test_tune_clip_graphed_matches_eager_and_counts_steps():
    """BASELINE configs[3] loop (training/video_swap_ft_coach.py:242-317) on a small Net3: `steps` passes over the clip, one optimiser step
    per frame, eroded region maps, foreground-weighted L2.  The graph-replayed loop (first two steps eager, then one captured step replayed
    with each frame's inputs) walks the trajectory of the plain eager loop, and the optimiser has taken exactly steps x frames steps."""
    import types
    install_dropin()
    from models.networks import Net3
    from e4s2024_amd import pti, ops
    opts = types.SimpleNamespace(fsencoder_type="psp", remaining_layer_idx=5, num_seg_cls=12, out_size=64, train_G=True,
                                 start_from_latent_avg=True, learn_in_w=False)
    n = 3
    vec = T(seeded.seeded_array(41, "vecs", (n, 12, 1280), dist="normal")).to(DEV)
    lab = T(seeded.blocky_labels(3, n, 12, 64, 8)).to(DEV)
    img = torch.tanh(T(seeded.seeded_array(5, "imgs", (n, 3, 64, 64), dist="normal"))).to(DEV)

    def make():
        net = Net3(opts)
        seeded.apply_seeded(net, 4, "net3")
        net = net.to(DEV).train()
        net.latent_avg = seeded.seeded_latent_avg(2, 10).to(DEV)
        return net

    maps, fg = pti.prepare_clip(lab, 1, (64, 64))
    assert tuple(fg.shape) == (n, 1, 64, 64) and np.array_equal(maps[1].cpu().numpy(), O.erode_mask(lab[1].cpu().numpy(), 1))
    hist = {}
    for graphed in (False, True):
        net = make()
        opt = torch.optim.Adam(pti.trainable_parameters(net), lr=1e-3, capturable=True, fused=True)
        hist[graphed] = pti.tune_clip(net, opt, img, lab, vec, steps=3, erode_radius=1, graphed=graphed, randomize_noise=False)
        taken = {int(st["step"].item()) for st in opt.state.values() if "step" in st}
        assert taken == {3 * n}, taken
    assert hist[False][-1] < hist[False][0]
    assert np.allclose(hist[True], hist[False], rtol=3e-3), hist


@pytest.mark.parametrize("a_kc,b_kc", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("M,N,K,batch,share_a", [(128, 128, 64, 1, False), (3, 288, 4096, 2, False), (200, 70, 100, 3, True), (513, 129, 36, 1, False),
                                                 (128, 1152, 65536, 1, False), (32, 288, 16, 4, True)])
def test_gemm_sb_against_fp64(a_kc, b_kc, M, N, K, batch, share_a):
    """``e4s_gemm_sb`` (csrc/gemm_sb.hip), every operand layout, ragged tiles, a K tail, a shared A, the split of a long K: against
    torch.matmul in float64, and bit-identical from run to run (the split partial sums are added in a fixed order)."""
    from e4s2024_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(1 if share_a else batch, M, K, generator=g)
    b = torch.randn(batch, K, N, generator=g)
    ref = torch.matmul(a.double(), b.double())
    a_dev = (a if a_kc else a.transpose(1, 2).contiguous()).to(DEV)
    b_dev = (b.transpose(1, 2).contiguous() if b_kc else b).to(DEV)
    c1 = ops.gemm_sb(a_dev, b_dev, a_kc, b_kc)
    c2 = ops.gemm_sb(a_dev, b_dev, a_kc, b_kc)
    torch.cuda.synchronize()
    assert tuple(c1.shape) == (batch, M, N) and torch.equal(c1, c2)
    err = (c1.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    record_parity(f"gemm_sb.{'kc' if a_kc else 'mc'}_{'kc' if b_kc else 'mc'}.{M}x{N}x{K}.rel_vs_fp64", err, 3e-5)
    assert err <= 3e-5


@pytest.mark.parametrize("cin,cout,h,ks,up,masked,bs", [(32, 48, 16, 3, 1, True, 2), (20, 130, 32, 3, 2, True, 1), (64, 3, 32, 1, 1, True, 2), (16, 32, 48, 3, 1, False, 2),
                                                    (144, 64, 16, 3, 1, True, 1)])
def test_implicit_weight_gradient_against_unfold_and_fp64(cin, cout, h, ks, up, masked, bs):
    """``e4s_mconv_wgrad`` (the modulated im2col rows produced inside the GEMM) == ``e4s_mconv_unfold`` followed by the explicit GEMM, and
    both agree with a float64 evaluation: plain / masked, 3x3 / 1x1, the composed up-sampling form (labels at the output resolution, four
    parities), labels outside the region range, ragged channel counts (tiles that straddle input channels)."""
    from e4s2024_amd import ops
    nreg = 5 if masked else 1
    g = torch.Generator().manual_seed(cin * 3 + cout + h)
    x = torch.randn(bs, cin, h, h, generator=g).to(DEV)
    G = up * up
    gz = torch.randn(G, bs, cout, h * h, generator=g).to(DEV)
    s = (torch.randn(bs, nreg, cin, generator=g) if masked else None)
    lab = None
    if masked:
        lab = torch.randint(0, nreg + 1, (bs, up * h, up * h), generator=g).to(torch.uint8)       # value nreg = "no region"
        lab[:, : up * h // 4] = 2                                                                  # a coherent part as well
        lab, s = lab.to(DEV), s.to(DEV)
    got = ops.mconv_wgrad(gz, x, s, lab, cout, ks, up)
    again = ops.mconv_wgrad(gz, x, s, lab, cout, ks, up)
    if masked:
        cols = ops._mconv_unfold(x, s, lab, ks, up)                                                # [G, bs, cin*KK, P]
    else:
        cols = ops.unfold2d(x, ks, 1, ks // 2, h, h)[None]
    ref64 = torch.matmul(gz.double(), cols.double().transpose(-1, -2)).reshape(G * bs, cout, cin * ks * ks)
    torch.cuda.synchronize()
    assert torch.equal(got, again)
    err = (got.double() - ref64).abs().max().item() / ref64.abs().max().item()
    record_parity(f"mconv_wgrad.cin{cin}_cout{cout}_h{h}_k{ks}_up{up}_{'masked' if masked else 'plain'}.rel_vs_fp64", err, 3e-5)
    assert err <= 3e-5


@pytest.mark.parametrize("bs,n,dim_in,dim_style,layers", [(1, 12, 1280, 512, 13), (2, 3, 40, 32, 2), (8, 5, 64, 128, 1)])
def test_local_mlps_native_backward_against_autograd(bs, n, dim_in, dim_style, layers):
    """The per-region LocalMLP stack: gradients of the input and of all 4 n parameters from ``e4s_grouped_linear_bwd`` (``ops._LocalMLPsGrad``)
    against autograd through the stock-PyTorch form (``torch_ref.local_mlps``), at the real size (12 x [512, 1280] + 12 x [6656, 512]) and small ones."""
    install_dropin()
    from models import networks
    from e4s2024_amd import ops, torch_ref
    torch.manual_seed(bs * 100 + n)
    mlps = [networks.LocalMLP(dim_in, dim_style, layers).to(DEV) for _ in range(n)]
    for m in mlps:
        for p in m.parameters():
            p.data.normal_(0, 0.5)
    x = torch.randn(bs, n, dim_in, device=DEV, requires_grad=True)
    add = torch.randn(layers * dim_style, device=DEV)
    wgt = torch.randn(bs, n, layers * dim_style, device=DEV)
    assert ops.NATIVE_BWD
    out = networks.local_mlps(mlps, x, addend=add)
    assert "LocalMLPsGrad" in type(out.grad_fn).__name__
    (out * wgt).sum().backward()
    got = [x.grad.clone()] + [p.grad.clone() for m in mlps for p in m.parameters()]
    x.grad = None
    for m in mlps:
        for p in m.parameters():
            p.grad = None
    l0, l2 = [m.mlp[0] for m in mlps], [m.mlp[2] for m in mlps]
    ref = torch_ref.local_mlps(x, [l.weight for l in l0], [l.bias for l in l0], [l.weight for l in l2], [l.bias for l in l2], l0[0].scale, l2[0].scale,
                               l0[0].lr_mul, l2[0].lr_mul, mlps[0].mlp[1].negative_slope, add)
    assert (out - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    (ref * wgt).sum().backward()
    want = [x.grad] + [p.grad for m in mlps for p in m.parameters()]
    worst = 0.0
    for a, b in zip(got, want):
        assert a.shape == b.shape
        worst = max(worst, (a - b).abs().max().item() / max(1e-6, b.abs().max().item()))
    record_parity(f"local_mlps.native_backward.bs{bs}_n{n}_{dim_in}x{dim_style}x{layers}.rel_vs_autograd", worst, 2e-4)
    assert worst <= 2e-4


@pytest.mark.parametrize("cin,cout,h,w,up,bs", [(32, 32, 16, 32, 1, 2), (48, 24, 20, 64, 1, 1), (64, 40, 14, 96, 2, 1), (40, 16, 32, 32, 2, 2), (32, 64, 64, 128, 1, 1)])
def test_fused_data_and_style_gradient_against_fp64_and_the_unfused_pair(cin, cout, h, w, up, bs):
    """``e4s_mconv_dgrad`` (U = Wᵀ gz kept in accumulators, modulated by the source position's region, summed with its shifts in LDS) against a
    float64 evaluation of the same sums and against ``e4s_gemm_sb`` + ``e4s_mconv_fold``: 32-wide maps (one tile column that owns every
    column), widths that are no multiple of the 30 owned columns, heights that are no multiple of the 6 owned rows, ragged channel counts,
    the composed up-sampling form (four parities, labels at the output resolution), labels outside the region range; bit-identical reruns."""
    from e4s2024_amd import ops
    nreg, G = 5, up * up
    g = torch.Generator().manual_seed(cin + 7 * cout + h + w)
    x = torch.randn(bs, cin, h, w, generator=g).to(DEV)
    gz = torch.randn(G, bs, cout, h * w, generator=g).to(DEV)
    wg = (torch.randn(G, cout, cin, 3, 3, generator=g) / (3 * cout ** 0.5)).to(DEV)
    s = torch.randn(bs, nreg, cin, generator=g).to(DEV)
    lab = torch.randint(0, nreg + 1, (bs, up * h, up * w), generator=g).to(torch.uint8)           # value nreg = "no region"
    lab[:, : up * h // 3] = 2
    lab[:, :, up * w // 2:] = torch.where(lab[:, :, up * w // 2:] == 1, torch.tensor(3, dtype=torch.uint8), lab[:, :, up * w // 2:])
    lab = lab.to(DEV)

    def run(fused):
        old = ops.DGRAD_FUSED
        ops.DGRAD_FUSED = fused
        try:
            return ops._mconv_input_grads(gz, wg, x, s, lab, up, True, True, False)[:2]
        finally:
            ops.DGRAD_FUSED = old
    dx, ds = run(True)
    dx2, ds2 = run(True)
    dxu, dsu = run(False)
    # float64
    U = torch.einsum("goik,gbop->gbikp", wg.double().view(G, cout, cin, 9), gz.double()).view(G, bs, cin, 9, h, w)
    xp = torch.nn.functional.pad(x.double(), (1, 1, 1, 1))
    dx64 = torch.zeros(bs, cin, h + 2, w + 2, dtype=torch.float64, device=DEV)
    ds64 = torch.zeros(bs, nreg, cin, dtype=torch.float64, device=DEV)
    s64 = torch.cat([s.double(), torch.zeros(bs, 1, cin, dtype=torch.float64, device=DEV)], 1)        # row nreg = no region
    for gi in range(G):
        ga, gb = gi // up, gi % up
        c = lab[:, ga::up, gb::up].long().clamp(max=nreg)                                              # [bs, h, w]
        smod = torch.gather(s64, 1, c.view(bs, -1, 1).expand(-1, -1, cin)).view(bs, h, w, cin).permute(0, 3, 1, 2)
        onehot = torch.nn.functional.one_hot(c, nreg + 1)[..., :nreg].double()                         # [bs, h, w, nreg]
        for ky in range(3):
            for kx in range(3):
                Uk = U[gi, :, :, ky * 3 + kx]                                                          # [bs, cin, h, w] at source positions p
                dx64[:, :, ky:ky + h, kx:kx + w] += Uk * smod                                          # lands on q = p + k - 1 (padded by one)
                ds64 += torch.einsum("bihw,bhwr->bri", Uk * xp[:, :, ky:ky + h, kx:kx + w], onehot)
    dx64 = dx64[:, :, 1:-1, 1:-1]
    torch.cuda.synchronize()
    assert torch.equal(dx, dx2) and torch.equal(ds, ds2)
    e_dx = (dx.double() - dx64).abs().max().item() / dx64.abs().max().item()
    e_ds = (ds.double() - ds64).abs().max().item() / ds64.abs().max().item()
    e_dxu = (dxu.double() - dx64).abs().max().item() / dx64.abs().max().item()
    e_dsu = (dsu.double() - ds64).abs().max().item() / ds64.abs().max().item()
    tag = f"mconv_dgrad.cin{cin}_cout{cout}_{h}x{w}_up{up}"
    record_parity(tag + ".dx_rel_vs_fp64", e_dx, 3e-5)
    record_parity(tag + ".ds_rel_vs_fp64", e_ds, 3e-5)
    assert e_dx <= 3e-5 and e_ds <= 3e-5, (e_dx, e_ds, e_dxu, e_dsu)
    assert e_dxu <= 3e-5 and e_dsu <= 3e-5


@pytest.mark.parametrize("co,ci", [(8, 40), (64, 33), (512, 512)])
def test_parity_composition_map_both_layouts_and_its_gradient(co, ci):
    """``e4s_small_map`` with T [36, 9] (the composition of an up layer's 3x3 weight with its blur into four parity weights,
    ``torch_ref._parity_weights``): the plain [36, N] layout and the grouped [4, N, 9] one the backward consumes, forward and gradient, against
    the einsum they replace; N = cout * cin not a multiple of the 256 columns of a workgroup."""
    from e4s2024_amd import ops, torch_ref
    g = torch.Generator().manual_seed(co + ci)
    ws = torch.randn(co, ci, 3, 3, generator=g).to(DEV).requires_grad_(True)
    blur = torch.tensor([1., 3., 3., 1.])
    blur = (blur[:, None] * blur[None, :] / blur.sum() ** 2 * 4).to(DEV)
    par = torch_ref._blur_shift(blur, torch.float32)[1]
    ref = torch.einsum("gyxkl,oikl->goiyx", par.double(), ws.double())
    gout = torch.randn(4, co, ci, 3, 3, generator=g).to(DEV)
    gref = torch.einsum("gyxkl,goiyx->oikl", par.double(), gout.double())
    plain = ops.small_map(ws.reshape(co, ci, 9), par.reshape(36, 9)).view(4, 3, 3, co, ci).permute(0, 3, 4, 1, 2)
    grouped = ops.small_map(ws.reshape(co, ci, 9), par.reshape(36, 9), grouped=True).view(4, co, ci, 3, 3)
    assert torch.equal(plain.contiguous(), grouped)
    assert (grouped.double() - ref).abs().max().item() <= 1e-6 * ref.abs().max().item()
    (gw,) = torch.autograd.grad(grouped, ws, gout)
    (gw_plain,) = torch.autograd.grad(plain, ws, gout)
    assert (gw - gw_plain).abs().max().item() <= 1e-6 * gw.abs().max().item()       # (same sums, the compiler contracts them differently)
    assert (gw.double() - gref).abs().max().item() <= 1e-6 * gref.abs().max().item()
    assert torch.equal(torch_ref._parity_weights(ws, blur, torch.float32), grouped)
