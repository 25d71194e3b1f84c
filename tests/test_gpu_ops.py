"""GPU parity of the stand-alone ops (SURVEY §8a rows a1, a2, a7 + the region-map helper) through the C ABI,
against the golden fixtures (reference outputs) and the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden, install_dropin
from e4s2024_amd import seeded
from oracle import e4s_oracle as O
from oracle import native as ON

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731


def maxdiff(a, b):
    return (a.detach().double().cpu() - torch.as_tensor(b).double()).abs().max().item()


@pytest.fixture(scope="module")
def op():
    install_dropin()
    import models.stylegan2.op as op
    return op


def test_native_library_is_loaded(op):
    from e4s2024_amd._lib import lib
    assert lib().cdll.e4s_abi_version() == 1
    with open("/proc/self/maps") as f:
        assert "libe4s_hip.so" in f.read()


def test_fused_leaky_relu_golden_fwd_bwd(op):
    g = load_golden("g1_fused_act")
    x = T(g["x"]).to(DEV).requires_grad_(True)
    b = T(g["bias"]).to(DEV).requires_grad_(True)
    y = op.fused_leaky_relu(x, b)
    assert maxdiff(y, g["y"]) <= 1e-6
    y.backward(T(g["grad_out"]).to(DEV))
    assert maxdiff(x.grad, g["grad_in"]) <= 1e-6
    assert maxdiff(b.grad, g["grad_bias"]) <= 1e-5
    y2 = op.fused_leaky_relu(T(g["x2d"]).to(DEV), T(g["b2d"]).to(DEV))
    assert maxdiff(y2, g["y2d"]) <= 1e-6
    m = op.FusedLeakyReLU(8).to(DEV)
    with torch.no_grad():
        m.bias.copy_(T(g["bias"]))
    assert maxdiff(m(T(g["x"]).to(DEV)), g["y"]) <= 1e-6


@pytest.mark.parametrize("shape", [(0, 4, 3, 3), (1, 1, 1, 1), (3, 5, 7, 9), (2, 32, 64, 64), (1, 3, 1023, 517)])
def test_fused_leaky_relu_shapes_vs_c_oracle(op, shape):
    x = seeded.seeded_array(31, "fa.x", shape, dist="normal")
    b = seeded.seeded_array(31, "fa.b", (shape[1],), dist="normal")
    y = op.fused_leaky_relu(T(x).to(DEV), T(b).to(DEV))
    assert tuple(y.shape) == shape
    if x.size:
        assert maxdiff(y, ON.fused_bias_act(x, b, None, 3, 0, 0.2, 2 ** 0.5)) <= 1e-6


def test_fused_bias_act_rejects_cpu_and_half(op):
    with pytest.raises(RuntimeError):
        op.fused_leaky_relu(torch.zeros(2, 3), torch.zeros(3))
    with pytest.raises(TypeError):
        op.fused_leaky_relu(torch.zeros(2, 3, device=DEV, dtype=torch.float16), torch.zeros(3, device=DEV, dtype=torch.float16))


def test_upfirdn2d_golden(op):
    g = load_golden("g2_upfirdn2d")
    for name in g["names"]:
        up, down, p0, p1 = (int(v) for v in g[f"{name}.p"])
        y = op.upfirdn2d(T(g[f"{name}.x"]).to(DEV), T(g[f"{name}.k"]).to(DEV), up=up, down=down, pad=(p0, p1))
        assert tuple(y.shape) == g[f"{name}.y"].shape, name
        assert maxdiff(y, g[f"{name}.y"]) <= 2e-6, name


@pytest.mark.parametrize("shape,up,down,pad", [((2, 3, 64, 64), 2, 1, (2, 1)), ((1, 32, 129, 129), 1, 1, (1, 1)),
                                               ((1, 2, 100, 37), 1, 2, (1, 1)), ((1, 1, 1, 1), 2, 1, (2, 1))])
def test_upfirdn2d_vs_c_oracle_and_backward(op, shape, up, down, pad):
    k = O.make_blur_kernel((1, 3, 3, 1), up * up)
    x = seeded.seeded_array(32, "ufd.x", shape, dist="normal")
    xg = T(x).to(DEV).requires_grad_(True)
    y = op.upfirdn2d(xg, k.to(DEV), up=up, down=down, pad=pad)
    ref = ON.upfirdn2d(x, k.numpy(), (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
    assert maxdiff(y, ref) <= 2e-6
    go = seeded.seeded_array(32, "ufd.go", tuple(y.shape), dist="normal")
    y.backward(T(go).to(DEV))
    with torch.enable_grad():
        xc = T(x).requires_grad_(True)
        O.upfirdn2d(xc, k, up, down, pad).backward(T(go))
    assert maxdiff(xg.grad, xc.grad) <= 5e-6


def test_grouped_linear_vs_oracle():
    from e4s2024_amd import ops
    for bs, groups, ind, outd, act in ((1, 12, 1280, 512, 1), (3, 12, 512, 6656, 0), (9, 1, 512, 512, 2), (2, 2, 30, 7, 0)):
        x = T(seeded.seeded_array(33, f"gl.x{ind}", (bs, groups, ind), dist="normal"))
        ws = [T(seeded.seeded_array(33, f"gl.w{g}.{ind}", (outd, ind))) for g in range(groups)]
        bs_ = [T(seeded.seeded_array(33, f"gl.b{g}.{ind}", (outd,), 0, 0.1)) for g in range(groups)]
        add = T(seeded.seeded_array(33, f"gl.a.{ind}", (outd,)))
        scale = 1 / np.sqrt(ind)
        out = ops.grouped_linear(x.to(DEV), [w.to(DEV) for w in ws], [b.to(DEV) for b in bs_], scale=scale, act=act, slope=0.01 if act == 1 else 0.2,
                                 addend=add.to(DEV))
        for g in range(groups):
            r = O.equal_linear(x[:, g], ws[g], bs_[g])
            if act == 1:
                r = torch.nn.functional.leaky_relu(r, 0.01)
            if act == 2:
                r = O.fused_leaky_relu(O.equal_linear(x[:, g], ws[g], None), bs_[g])
            assert maxdiff(out[:, g], r + add) <= 2e-5 * max(1.0, r.abs().max().item())


def test_mask_to_labels():
    from e4s2024_amd import ops
    lab = seeded.iid_labels(5, 2, 12, 64)
    mask = seeded.labels_to_onehot(lab, 12)
    mask[0, :, 3, 5] = 0                                  # a pixel with no class -> LABEL_NONE
    got = ops.mask_to_labels(mask.to(DEV)).cpu().numpy()
    exp = lab.copy()
    exp[0, 3, 5] = 255
    assert (got == exp).all()
    bad = mask.clone()
    bad[1, 4, 0, 0] = 0.5
    with pytest.raises(ValueError):
        ops.mask_to_labels(bad.to(DEV))
    multi = mask.clone()
    multi[1, :2, 1, 1] = 1.0
    with pytest.raises(ValueError):
        ops.mask_to_labels(multi.to(DEV))
