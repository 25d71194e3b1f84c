"""Drop-in for the reference's ``swap_face_fine/face_parsing/resnet.py`` (``Resnet18`` :58-99, ``BasicBlock`` :21-49).

Same module/parameter names (191-key BiSeNet state_dict loads unchanged).  Every ``conv -> BatchNorm2d(eval) [-> ReLU]``
is one launch of the implicit-GEMM conv kernel with the BN folded into weights+bias and the residual add / ReLU in its
epilogue.  Nothing is downloaded at construction (the reference fetches ImageNet weights in ``init_weight`` :83-90; the
face-parsing checkpoint ``79999_iter.pth`` overwrites them anyway)."""
import torch
import torch.nn as nn

from e4s2024_amd import ops


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def _eval_only(m):
    if m.training:
        raise RuntimeError(f"{type(m).__name__}: the MI355X parser path folds BatchNorm running statistics and runs in eval() mode only "
                           "(training BiSeNet is out of scope)")


class BasicBlock(nn.Module):
    def __init__(self, in_chan, out_chan, stride=1):
        super(BasicBlock, self).__init__()
        self.conv1 = conv3x3(in_chan, out_chan, stride)
        self.bn1 = nn.BatchNorm2d(out_chan)
        self.conv2 = conv3x3(out_chan, out_chan)
        self.bn2 = nn.BatchNorm2d(out_chan)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        self.stride = stride
        if in_chan != out_chan or stride != 1:
            self.downsample = nn.Sequential(nn.Conv2d(in_chan, out_chan, kernel_size=1, stride=stride, bias=False), nn.BatchNorm2d(out_chan))
        self._w = [ops.PreparedConv(exact=ops.PARSER_EXACT) for _ in range(3)]

    def forward(self, x):
        _eval_only(self)
        residual = ops.conv2d(x, self._w[0].get(self.conv1.weight, self.bn1), self.stride, 1, relu=True)
        shortcut = x
        if self.downsample is not None:
            shortcut = ops.conv2d(x, self._w[2].get(self.downsample[0].weight, self.downsample[1]), self.stride, 0)
        # relu(shortcut + bn2(conv2(residual)))  (reference :46-48), add and ReLU fused into the conv epilogue
        return ops.conv2d(residual, self._w[1].get(self.conv2.weight, self.bn2), 1, 1, residual=shortcut, relu=True)


def create_layer_basic(in_chan, out_chan, bnum, stride=1):
    layers = [BasicBlock(in_chan, out_chan, stride=stride)]
    for _ in range(bnum - 1):
        layers.append(BasicBlock(out_chan, out_chan, stride=1))
    return nn.Sequential(*layers)


class Resnet18(nn.Module):
    def __init__(self):
        super(Resnet18, self).__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = create_layer_basic(64, 64, bnum=2, stride=1)
        self.layer2 = create_layer_basic(64, 128, bnum=2, stride=2)
        self.layer3 = create_layer_basic(128, 256, bnum=2, stride=2)
        self.layer4 = create_layer_basic(256, 512, bnum=2, stride=2)
        self._w = ops.PreparedConv(exact=ops.PARSER_EXACT)

    def forward(self, x):
        _eval_only(self)
        x = ops.conv2d(x, self._w.get(self.conv1.weight, self.bn1), 2, 3, relu=True)
        x = ops.maxpool3x3s2(x)
        x = self.layer1(x)
        feat8 = self.layer2(x)       # 1/8
        feat16 = self.layer3(feat8)  # 1/16
        feat32 = self.layer4(feat16)  # 1/32
        return feat8, feat16, feat32

    def init_weight(self):
        """The reference downloads torchvision's resnet18 here (:83-90); no network in this build, and the parser checkpoint
        replaces every one of these tensors."""
        return None

    def get_params(self):
        wd_params, nowd_params = [], []
        for _, module in self.named_modules():
            if isinstance(module, (nn.Linear, nn.Conv2d)):
                wd_params.append(module.weight)
                if module.bias is not None:
                    nowd_params.append(module.bias)
            elif isinstance(module, nn.BatchNorm2d):
                nowd_params += list(module.parameters())
        return wd_params, nowd_params
