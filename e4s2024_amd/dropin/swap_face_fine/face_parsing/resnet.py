"""Drop-in for the reference's ``swap_face_fine/face_parsing/resnet.py`` (``Resnet18`` :58-99, ``BasicBlock`` :21-49).

Same module/parameter names (191-key BiSeNet state_dict loads unchanged).  Every ``conv -> BatchNorm2d(eval) [-> ReLU]``
is one launch of the implicit-GEMM conv kernel with the BN folded into weights+bias and the residual add / ReLU in its
epilogue.  Nothing is downloaded at construction (the reference fetches ImageNet weights in ``init_weight`` :83-90; the
face-parsing checkpoint ``79999_iter.pth`` overwrites them anyway)."""
import torch
import torch.nn as nn

from e4s2024_amd import ops


def _conv(cin, cout, k, stride):
    return nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False)


def conv3x3(in_planes, out_planes, stride=1):      # public name of the reference module (:14-17)
    return _conv(in_planes, out_planes, 3, stride)


def _eval_only(m):
    if m.training:
        raise RuntimeError(f"{type(m).__name__}: the MI355X parser path folds BatchNorm running statistics and runs in eval() mode only "
                           "(training BiSeNet is out of scope)")


class BasicBlock(nn.Module):
    """Two 3x3 conv+BN stages with an identity or 1x1-conv shortcut; registration order = the reference's state_dict order."""

    def __init__(self, in_chan, out_chan, stride=1):
        super().__init__()
        self.stride = stride
        for name, mod in (("conv1", _conv(in_chan, out_chan, 3, stride)), ("bn1", nn.BatchNorm2d(out_chan)),
                          ("conv2", _conv(out_chan, out_chan, 3, 1)), ("bn2", nn.BatchNorm2d(out_chan)), ("relu", nn.ReLU(inplace=True))):
            self.add_module(name, mod)
        projected = stride != 1 or in_chan != out_chan
        self.downsample = nn.Sequential(_conv(in_chan, out_chan, 1, stride), nn.BatchNorm2d(out_chan)) if projected else None
        self._w = [ops.PreparedConv(exact=ops.PARSER_EXACT) for _ in range(3)]

    def forward(self, x):
        _eval_only(self)
        branch = ops.conv2d(x, self._w[0].get(self.conv1.weight, self.bn1), self.stride, 1, relu=True)
        shortcut = x if self.downsample is None else ops.conv2d(x, self._w[2].get(self.downsample[0].weight, self.downsample[1]), self.stride, 0)
        # relu(shortcut + bn2(conv2(branch)))  (reference :46-48): the add and the ReLU run in the second conv's epilogue
        return ops.conv2d(branch, self._w[1].get(self.conv2.weight, self.bn2), 1, 1, residual=shortcut, relu=True)


def create_layer_basic(in_chan, out_chan, bnum, stride=1):
    return nn.Sequential(*[BasicBlock(in_chan if i == 0 else out_chan, out_chan, stride=stride if i == 0 else 1) for i in range(bnum)])


class Resnet18(nn.Module):
    STAGES = ((64, 64, 1), (64, 128, 2), (128, 256, 2), (256, 512, 2))     # (in, out, stride) of layer1..layer4, two blocks each

    def __init__(self):
        super().__init__()
        self.conv1 = _conv(3, 64, 7, 2)
        self.bn1 = nn.BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        for i, (cin, cout, stride) in enumerate(self.STAGES, start=1):
            setattr(self, f"layer{i}", create_layer_basic(cin, cout, bnum=2, stride=stride))
        self._w = ops.PreparedConv(exact=ops.PARSER_EXACT)

    def forward(self, x):
        _eval_only(self)
        stem = ops.maxpool3x3s2(ops.conv2d(x, self._w.get(self.conv1.weight, self.bn1), 2, 3, relu=True))
        feat8 = self.layer2(self.layer1(stem))      # 1/8 of the input resolution
        feat16 = self.layer3(feat8)                 # 1/16
        return feat8, feat16, self.layer4(feat16)   # 1/32

    def init_weight(self):
        """The reference downloads torchvision's resnet18 here (:83-90); no network in this build, and the parser checkpoint
        replaces every one of these tensors."""
        return None

    def get_params(self):
        """(weights with decay, parameters without): conv / linear weights vs their biases and all BatchNorm parameters (:92-99)."""
        decay, no_decay = [], []
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                no_decay.extend(m.parameters())
            elif isinstance(m, (nn.Conv2d, nn.Linear)):
                decay.append(m.weight)
                if m.bias is not None:
                    no_decay.append(m.bias)
        return decay, no_decay
