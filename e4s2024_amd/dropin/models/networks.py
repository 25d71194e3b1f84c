"""Drop-in for the reference's ``models/networks.py``: ``LocalMLP`` (:23-49) and ``Net3`` (:51-277).

Same constructor (``Net3(opts)``), sub-module names (``encoder``, ``MLPs``, ``G``), state_dict keys, method names, argument
order and return tuples.  ``latent_avg`` is a plain attribute assigned by the caller after construction, exactly as the
reference's loaders do (face_swap_video_pipeline.py:557-561).  The 12 LocalMLPs run as ONE grouped launch per layer
(``e4s_grouped_linear``) instead of 24 small GEMVs.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from e4s2024_amd import ops, torch_ref
from models.stylegan2.model import EqualLinear, Generator
from models.encoders.psp_encoders import FSEncoder_PSP


class LocalMLP(nn.Module):
    """reference :23-49: EqualLinear(dim_component, dim_style) -> LeakyReLU(0.01) -> EqualLinear(dim_style, dim_style*num_w_layers)."""

    def __init__(self, dim_component=512, dim_style=512, num_w_layers=18, latent_squeeze_ratio=1):
        super(LocalMLP, self).__init__()
        self.dim_component = dim_component
        self.dim_style = dim_style
        self.num_w_layers = num_w_layers
        self.mlp = nn.Sequential(
            EqualLinear(dim_component, dim_style // latent_squeeze_ratio, lr_mul=1),
            nn.LeakyReLU(),
            EqualLinear(dim_style // latent_squeeze_ratio, dim_style * num_w_layers, lr_mul=1),
        )

    def forward(self, x):
        return local_mlps([self], x[:, None, :]).view(-1, self.num_w_layers, self.dim_style)


def local_mlps(mlps, x, addend=None):
    """All ``len(mlps)`` LocalMLPs on ``x [bs, n, dim_component]`` in two grouped launches → ``[bs, n, num_w_layers*dim_style]``.
    ``addend`` (optional ``[num_w_layers*dim_style]``) is added to every row in the second launch's epilogue."""
    l0 = [m.mlp[0] for m in mlps]
    l2 = [m.mlp[2] for m in mlps]
    slope = mlps[0].mlp[1].negative_slope
    h = ops.grouped_linear(x, [l.weight for l in l0], [l.bias for l in l0], scale=l0[0].scale, bias_mul=l0[0].lr_mul, act=1, slope=slope)
    out = ops.grouped_linear(h, [l.weight for l in l2], [l.bias for l in l2], scale=l2[0].scale, bias_mul=l2[0].lr_mul, act=0, addend=addend)
    deps = [x] + [p for l in l0 + l2 for p in (l.weight, l.bias)]
    n = len(mlps)
    if (torch.is_grad_enabled() and ops.NATIVE_BWD and any(d.requires_grad for d in deps) and x.shape[0] <= 8 and x.shape[2] % 4 == 0
            and h.shape[2] % 4 == 0 and x.is_contiguous() and all(l.bias is not None for l in l0 + l2)):
        # gradients from the grouped-linear backward kernels (no re-evaluation, no stacked copy of the weights)
        return ops.local_mlps_grad(x, out, h, [l.weight for l in l0], [l.bias for l in l0], [l.weight for l in l2], [l.bias for l in l2],
                                   l0[0].scale, l2[0].scale, l0[0].lr_mul, l2[0].lr_mul, slope)

    def ref(xr, *ps):     # ps: (w, b) of the n first layers, then of the n second layers
        return torch_ref.local_mlps(xr, ps[0:2 * n:2], ps[1:2 * n:2], ps[2 * n::2], ps[2 * n + 1::2], l0[0].scale, l2[0].scale, l0[0].lr_mul,
                                    l2[0].lr_mul, slope, addend)
    return ops._attach("LocalMLP", out, *deps, ref=ref)


class Net3(nn.Module):
    """FSEncoder + per-region MLPs + masked StyleGAN2 (reference :51-277)."""

    def __init__(self, opts):
        super(Net3, self).__init__()
        self.opts = opts
        assert self.opts.fsencoder_type in ["psp", "sean"]
        if self.opts.fsencoder_type == "psp":
            self.encoder = FSEncoder_PSP(mode='ir_se', opts=self.opts)
            dim_s_code = 256 + 512 + 512
        else:
            raise NotImplementedError("fsencoder_type='sean' is unused by the default options "
                                      "(options/our_swap_face_pipeline_options.py:17) and out of scope")
        self.split_layer_idx = 5
        self.remaining_layer_idx = self.opts.remaining_layer_idx

        self.MLPs = nn.ModuleList()
        for i in range(self.opts.num_seg_cls):
            self.MLPs.append(LocalMLP(dim_component=dim_s_code, dim_style=512,
                                      num_w_layers=self.remaining_layer_idx if self.remaining_layer_idx != 17 else 18))

        self.G = Generator(size=self.opts.out_size, style_dim=512, n_mlp=8, split_layer_idx=self.split_layer_idx,
                           remaining_layer_idx=self.remaining_layer_idx)

        # reference :82-95
        if not self.opts.train_G:
            for param in self.G.parameters():
                param.requires_grad = False
        else:
            for param in self.G.style.parameters():
                param.requires_grad = False
        if self.remaining_layer_idx != 17:
            for param in self.G.convs[-(17 - self.remaining_layer_idx):].parameters():
                param.requires_grad = False
            for param in self.G.to_rgbs[-(17 - self.remaining_layer_idx) // 2 - 1:].parameters():
                param.requires_grad = False

    # ------------------------------------------------------------------ pieces shared by the public methods
    def _encode(self, img, mask):
        # F.interpolate(img, (256, 256), mode='bilinear')  (reference :217) -> device kernel, align_corners=False, no antialias
        return self.encoder(ops.bilinear_resize(img, (256, 256), align_corners=False), mask)

    def _codes_from_vectors(self, style_vectors):
        rli = self.remaining_layer_idx
        bs, num_comp = style_vectors.size(0), style_vectors.size(1)
        nw = self.MLPs[0].num_w_layers
        if not self.opts.start_from_latent_avg:
            # the reference leaves style_codes undefined on this branch (:239-253 only assign inside the if)
            raise NotImplementedError("start_from_latent_avg=False is not a supported configuration of the reference either")
        latent_avg = self.latent_avg.to(device=style_vectors.device, dtype=torch.float32)
        if self.opts.learn_in_w:
            raise NotImplementedError("learn_in_w is off in every live configuration (options/our_swap_face_pipeline_options.py:48)")
        # codes[b, c, :nw] = MLP_c(v[b, c]) + latent_avg[:nw]   (reference :247 / :251), fused as the second launch's addend
        codes = local_mlps(list(self.MLPs), style_vectors, addend=latent_avg[:nw].reshape(-1)).view(bs, num_comp, nw, 512)
        if rli != 17:
            remaining = latent_avg[rli:, :][None, None].expand(bs, num_comp, -1, -1)      # reference :248
            codes = torch.cat([codes, remaining], dim=2)
        return codes

    # ------------------------------------------------------------------ public API (reference :98-277)
    def forward(self, img, mask, resize=False, randomize_noise=True, return_latents=False):
        codes_vector, structure_feats = self._encode(img, mask)
        codes = self._codes_from_vectors(codes_vector)
        images1, result_latent, structure_feats_GT = self.G([codes], structure_feats, mask, input_is_latent=True,
                                                            randomize_noise=randomize_noise, return_latents=return_latents,
                                                            use_structure_code=False)
        if return_latents:
            return images1, structure_feats_GT, result_latent
        return images1, structure_feats_GT

    def get_style(self, img, mask):
        codes_vector, structure_feats = self._encode(img, mask)
        return structure_feats, self._codes_from_vectors(codes_vector)

    def get_style_vectors(self, img, mask):
        style_vectors, structure_feats = self._encode(img, mask)
        return style_vectors, structure_feats

    def cal_style_codes(self, style_vectors):
        return self._codes_from_vectors(style_vectors)

    def gen_img(self, struc_codes, style_codes, mask, randomize_noise=True, noise=None, return_latents=False):
        images, result_latent, structure_feats = self.G([style_codes], struc_codes, mask, input_is_latent=True,
                                                        randomize_noise=randomize_noise, noise=noise, return_latents=return_latents,
                                                        use_structure_code=False)
        if return_latents:
            return images, result_latent, structure_feats
        return images, -1, structure_feats


Net = Net3  # BASELINE.json's north_star calls it models.networks.Net; the reference only defines Net3 (SURVEY naming note)
