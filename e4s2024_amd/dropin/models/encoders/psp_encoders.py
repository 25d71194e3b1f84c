"""Drop-in for the regional-style encoder of the reference: ``FSEncoder_PSP`` (models/encoders/psp_encoders.py:319-401)
together with the unit it is built from, ``bottleneck_IR_SE_Ours`` / ``SEModule`` / ``get_block``
(models/encoders/helpers.py:19-24, 56-72, 122-144).  The reference's ``helpers.py`` itself is NOT overridden: its other
blocks belong to the ArcFace ID-loss network (criteria/, out of scope) and keep running on stock PyTorch.

Same constructor, state_dict keys and ``forward(x, segmap) -> (codes_vector [bs, n_cls, 1280], zeros [bs, 512, 16, 16])``.
The other encoders of that file (``GradualStyleEncoder``, ``Backbone*``, ``FSEncoder_SEAN``, :35-316) are unused with the
default ``--fsencoder_type psp`` and out of scope.

The torch.nn layers below are *parameter holders* that give the state_dict the reference's key names
(``res_layer.{1,2,3}.weight``, ``res_layer.5.fc{1,2}.weight``, ``shortcut_layer.0.weight``); the arithmetic runs in
libe4s_hip.so (conv.hip / conv_mx3.hip / norm.hip).  One unit (inference) = three launches, five with a shortcut convolution:

    conv3x3(IN(x) applied while staging, PReLU epilogue) -> conv3x3(stride)
             -> [shortcut conv1x1(stride) -> stats] -> IN statistics + IN apply * gate + shortcut add (+ the statistics of the result for the next unit)

(the SE gate behind the affine-free InstanceNorm is the constant 1/2 — ops.SE_GATE_IS_HALF; computed from the plane's measured mean when that is switched off)
"""
from collections import namedtuple

import torch
from torch.nn import Conv2d, PReLU, ReLU, Sigmoid, MaxPool2d, AdaptiveAvgPool2d, Sequential, Module, InstanceNorm2d

from e4s2024_amd import ops


class Bottleneck(namedtuple('Block', ['in_channel', 'depth', 'stride'])):
    """A named tuple describing a ResNet block."""


def get_block(in_channel, depth, num_units, stride=2):
    return [Bottleneck(in_channel, depth, stride)] + [Bottleneck(depth, depth, 1) for _ in range(num_units - 1)]


class SEModule(Module):
    """reference helpers.py:56-72 — squeeze-excite gate ``x * sigmoid(fc2(relu(fc1(mean(x)))))``, no biases."""

    def __init__(self, channels, reduction):
        super(SEModule, self).__init__()
        self.avg_pool = AdaptiveAvgPool2d(1)
        self.fc1 = Conv2d(channels, channels // reduction, kernel_size=1, padding=0, bias=False)
        self.relu = ReLU(inplace=True)
        self.fc2 = Conv2d(channels // reduction, channels, kernel_size=1, padding=0, bias=False)
        self.sigmoid = Sigmoid()

    def gate(self, pooled):
        """pooled ``[bs, C]`` -> gate ``[bs, C]`` (one launch; the hidden width of the reference's reduction 16 is at most 32)."""
        if self.fc1.weight.shape[0] <= 64:
            return ops.se_gate(pooled, self.fc1.weight, self.fc2.weight)
        return ops.vec_fc(ops.vec_fc(pooled, self.fc1.weight, act=ops.ACT_RELU), self.fc2.weight, act=ops.ACT_SIGMOID)

    def forward(self, x):
        out = ops.norm_gate_add(x, gate=self.gate(ops.plane_stats(x)))
        return ops._attach("SEModule", out, x, self.fc1.weight, self.fc2.weight)


class bottleneck_IR_SE_Ours(Module):
    """reference helpers.py:122-144."""

    def __init__(self, in_channel, depth, stride):
        super(bottleneck_IR_SE_Ours, self).__init__()
        self.stride = stride
        if in_channel == depth:
            self.shortcut_layer = MaxPool2d(1, stride)
        else:
            self.shortcut_layer = Sequential(Conv2d(in_channel, depth, (1, 1), stride, bias=False), InstanceNorm2d(depth))
        self.res_layer = Sequential(
            InstanceNorm2d(in_channel),
            Conv2d(in_channel, depth, (3, 3), (1, 1), 1, bias=False),
            PReLU(depth),
            Conv2d(depth, depth, (3, 3), stride, 1, bias=False),
            InstanceNorm2d(depth),
            SEModule(depth, 16),
        )
        self._w = [ops.PreparedConv() for _ in range(3)]
        # the two 3x3 convolutions' other prepared copies (stride 1 only): Winograd-domain weights, the DMA-fed kernels' slots
        self._wino = [(self._w[i], ops.PreparedWinograd(), ops.PreparedMx()) for i in range(2)]

    def forward(self, x):
        rl = self.res_layer
        # InstanceNorm2d(in_channel) statistics: the producer of x (the previous unit's / the input layer's norm_gate_add) computed them
        # in its own launch and left them on the tensor; anything else is measured here
        st = getattr(x, "_e4s_in_stats", None)
        if st is not None and st[2] == rl[0].eps and not torch.is_grad_enabled():
            mean, rstd = st[0], st[1]
        else:
            mean, rstd = ops.plane_stats(x, rl[0].eps)
        # (a stride-2 unit whose two convolutions both run on the two-phase kernel hands r over as phase planes: the second reads it coalesced)
        depth = rl[1].weight.shape[0]
        phased = (self.stride == 2 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and ops.conv3x3_s1_takes_mx3(x, depth)
                  and ops.conv3x3_s2_takes_mx(x.shape[0], depth, rl[3].weight.shape[0], x.shape[2], x.shape[3], x.device))
        # (and channel-blocked whenever both run there — ops.conv3x3_s1_c4_pair: the map between the two convolutions has no other reader, helpers.py:128-139)
        #  or, better, as the consumer's prepared operands — ops.ENC_PREP_LINK)
        link = (self.stride == 1 or phased) and ops.conv3x3_s1_c4_pair(x, depth, rl[3].weight.shape[0], self.stride)
        prep = link and ops.ENC_PREP_LINK and depth % 32 == 0 and (x.shape[2] * x.shape[3]) % 4 == 0
        r = ops.conv3x3_s1(x, rl[1].weight, self._wino[0], in_norm=(mean, rstd), prelu=rl[2].weight, out_phased=phased,      # direct kernel or Winograd (ops.winograd_route)
                           out_c4=link and not prep and ops.ENC_C4_LINK, out_prep=prep)
        if self.stride == 1:
            r = ops.conv3x3_s1(r, rl[3].weight, self._wino[1])
        else:
            r = ops.conv3x3_s2(r, rl[3].weight, self._wino[1]) if self.stride == 2 else ops.conv2d(r, self._w[1].get(rl[3].weight), self.stride, 1)
        # (the constant gate needs the reference's preconditions, checked on the modules as they are NOW: an affine-free InstanceNorm without running statistics in
        #  front of bias-free fc1 / fc2 — a variant or a checkpoint with an affine norm or biases gets the computed gate)
        se, nrm = rl[5], rl[4]
        half_ok = (not getattr(nrm, "affine", False) and not getattr(nrm, "track_running_stats", False)
                   and getattr(se.fc1, "bias", None) is None and getattr(se.fc2, "bias", None) is None)
        self_stats = ops.SE_GATE_IS_HALF and half_ok and not torch.is_grad_enabled()
        if self_stats:
            # SEModule behind an affine-free InstanceNorm: its squeeze is the mean of a normalised plane = 0, its bias-free gate sigmoid(0) = 1/2 (ops.SE_GATE_IS_HALF);
            # nothing then needs r's statistics before the unit's last launch, which computes them itself
            m2 = r2 = None
            gate = ops.half_gate(r.shape[0], r.shape[1], r.device)
        else:
            m2, r2, pooled = ops.plane_stats(r, rl[4].eps, want_nmean=True)             # IN statistics + mean of the normalised map
            gate = rl[5].gate(pooled)
        if isinstance(self.shortcut_layer, MaxPool2d):
            sc, sc_stats, sc_stride = x, None, self.stride                              # MaxPool2d(1, stride) == strided subsample
        else:
            sc = ops.conv2d(x, self._w[2].get(self.shortcut_layer[0].weight), self.stride, 0)
            sc_stats, sc_stride = ops.plane_stats(sc, self.shortcut_layer[1].eps), 1
        if torch.is_grad_enabled():
            out = ops.norm_gate_add(r, m2, r2, gate, sc, sc_stats, sc_stride)
            return ops._attach("bottleneck_IR_SE_Ours", out, x, *[p for p in self.parameters()])
        out, om, orr = ops.norm_gate_add(r, m2, r2, gate, sc, sc_stats, sc_stride, stats_eps=rl[0].eps, self_eps=rl[4].eps if self_stats else None)
        out._e4s_in_stats = (om, orr, rl[0].eps)        # every unit's first InstanceNorm has the default eps
        return out


class FSEncoder_PSP(Module):
    def __init__(self, mode='ir_se', opts=None):
        super(FSEncoder_PSP, self).__init__()
        if mode != 'ir_se':
            raise NotImplementedError("only mode='ir_se' (bottleneck_IR_SE_Ours) is used by Net3 (models/networks.py:59)")
        blocks = [
            get_block(in_channel=64, depth=128, num_units=3),
            get_block(in_channel=128, depth=256, num_units=4),
            get_block(in_channel=256, depth=512, num_units=14),
            get_block(in_channel=512, depth=512, num_units=3),
        ]
        self.n_styles = 11
        self.num_seg_cls = int(getattr(opts, "num_seg_cls", 12)) if opts is not None else 12
        self.input_layer = Sequential(Conv2d(3, 64, (3, 3), 1, 1, bias=False), InstanceNorm2d(64), PReLU(64))
        modules = []
        for block in blocks:
            for bottleneck in block:
                modules.append(bottleneck_IR_SE_Ours(bottleneck.in_channel, bottleneck.depth, bottleneck.stride))
        self.body = Sequential(*modules)
        self._w_in = ops.PreparedConv()

    def get_per_comp_styleCode(self, style_feats, segmap):
        """Masked average pooling per region (reference :355-375): the mask is sampled 'nearest' at the feature size; a region
        with no pixel gives a zero vector.  One launch, no host sync (the reference loops bs x n_cls times in Python)."""
        # segmap: one-hot float [bs, n_cls, H, W] (reference callers) or, engine extension, a uint8 region map [bs, H, W]
        nreg = self.num_seg_cls if (segmap.dtype == torch.uint8 and segmap.dim() == 3) else segmap.shape[1]
        return ops.masked_avg_pool(style_feats, ops.mask_to_labels(segmap), nreg)

    def forward(self, x, segmap):
        # (ops.guarded: the stride-1 3x3 convolutions may run in f16 + fp6 arithmetic; a pass that left the f16 range is re-run in split-bf16)
        return ops.guarded(lambda: self.codes(self.features(x), segmap))

    def features(self, x):
        """The feature maps the style codes are pooled from (after units 6, 20 and 23) — everything of ``forward`` that does not depend on the
        region map, so that a caller can run the face parser next to it (``pipeline.swap_batch``)."""
        il = self.input_layer
        y = ops.conv2d(x, self._w_in.get(il[0].weight), 1, 1)
        if torch.is_grad_enabled():
            mean, rstd = ops.plane_stats(y, il[1].eps)
            x = ops.norm_gate_add(y, mean, rstd, prelu=il[2].weight)
        else:       # InstanceNorm2d(64) statistics + apply + PReLU + the statistics of the result (the first unit's InstanceNorm) in one launch
            x, om, orr = ops.norm_gate_add(y, prelu=il[2].weight, stats_eps=il[1].eps, self_eps=il[1].eps)
            x._e4s_in_stats = (om, orr, il[1].eps)
        taps = {}
        for i, unit in enumerate(self.body):
            x = unit(x)
            if i in (6, 20, 23):
                taps[i] = x
        taps["last"] = x
        return taps

    def codes(self, taps, segmap):
        structure_feats = torch.zeros_like(taps["last"])                                # reference :392
        codes_vector = torch.cat([self.get_per_comp_styleCode(taps[6], segmap),
                                  self.get_per_comp_styleCode(taps[20], segmap),
                                  self.get_per_comp_styleCode(taps[23], segmap)], dim=2)
        return codes_vector, structure_feats
