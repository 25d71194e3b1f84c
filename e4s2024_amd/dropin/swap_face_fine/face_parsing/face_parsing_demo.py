"""Drop-in for the reference's ``swap_face_fine/face_parsing/face_parsing_demo.py``: ``BicubicDownSample`` (:15-84),
``FaceParser`` (:132-176), ``init_faceParsing_pretrained_model`` (:180-185), ``faceParsing_demo`` (:187-200),
``vis_parsing_maps`` (:87-129).

The pre-processing (bicubic /2, clamp, ImageNet normalise), BiSeNet, the bilinear up-sampling + argmax and the 19->12 label
remap (datasets/dataset.py:58-108) all run on the device; only the PIL image goes up and a uint8 label map comes down.
``FaceParser.parse_batch`` is the batched tensor entry the multi-GPU runner uses."""
import math

import numpy as np
import torch
from PIL import Image
from torch import nn

from e4s2024_amd import ops
from swap_face_fine.face_parsing.model import BiSeNet, seg_mean, seg_std

# datasets/dataset.py:58-108 — every label not listed (15 neck_l, 16 cloth, 18 hat) maps to 0
_REMAP_19_TO_12 = {0: 0, 12: 1, 13: 1, 2: 2, 3: 2, 4: 3, 5: 3, 17: 4, 10: 5, 1: 6, 7: 7, 8: 7, 14: 8, 11: 9, 6: 10, 9: 11}


def remap_lut():
    lut = np.zeros(256, dtype=np.uint8)
    for s, d in _REMAP_19_TO_12.items():
        lut[s] = d
    return lut


def _pil_to_tensor01(img, device):
    """torchvision.transforms.ToTensor()(img)[:3].unsqueeze(0).to(device) for an 8-bit PIL image."""
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(np.array(a, copy=True)).to(device)
    return (t.permute(2, 0, 1)[:3].float() / 255.0).unsqueeze(0).contiguous()


class BicubicDownSample(nn.Module):
    """reference :15-84 (``a = -0.5``, reflect padding, separable ``4*factor`` taps); factor 2 or 4 on the device kernel."""

    def bicubic_kernel(self, x, a=-0.50):
        abs_x = abs(float(x))
        if abs_x <= 1.:
            return (a + 2.) * abs_x ** 3 - (a + 3.) * abs_x ** 2 + 1
        elif 1. < abs_x < 2.:
            return a * abs_x ** 3 - 5. * a * abs_x ** 2 + 8. * a * abs_x - 4. * a
        return 0.0

    def __init__(self, factor=4, cuda=True, padding='reflect'):
        super().__init__()
        if padding != 'reflect':
            raise NotImplementedError("only reflect padding (the reference default) is built")
        self.factor = factor
        size = factor * 4
        k = torch.tensor([self.bicubic_kernel((i - math.floor(size / 2) + 0.5) / factor) for i in range(size)], dtype=torch.float32)
        self.register_buffer("taps", k / torch.sum(k), persistent=False)
        self.padding = padding

    def forward(self, x, nhwc=False, clip_round=False, byte_output=False):
        if nhwc or clip_round or byte_output:
            raise NotImplementedError("nhwc / clip_round / byte_output are not used on the parser path")
        return ops.bicubic_down_normalize(x, self.taps.to(x.device), self.factor)


class FaceParser(nn.Module):
    def __init__(self, seg_ckpt, size=1024, device="cuda"):
        super(FaceParser, self).__init__()
        self.seg_ckpt = seg_ckpt
        self.size = size
        self.device = device
        self.load_segmentation_network()
        self.load_downsampling()
        self.register_buffer("_mean", seg_mean.reshape(3).clone(), persistent=False)
        self.register_buffer("_std", seg_std.reshape(3).clone(), persistent=False)
        self.register_buffer("_lut12", torch.from_numpy(remap_lut()), persistent=False)
        self.to(device)

    def load_downsampling(self):
        self.downsample = BicubicDownSample(factor=self.size // 512)
        self.downsample_256 = BicubicDownSample(factor=self.size // 256)

    def load_segmentation_network(self):
        self.seg = BiSeNet(n_classes=19)
        self.seg.to(self.device)
        if self.seg_ckpt is not None:
            self.seg.load_state_dict(torch.load(self.seg_ckpt, map_location=self.device))
        for param in self.seg.parameters():
            param.requires_grad = False
        self.seg.eval()

    # ---- tensor entry points -------------------------------------------------------------------------------------
    def preprocess_tensor(self, img01, downsample=True, pm1=False, out=None):
        """``[bs, 3, S, S]`` in [0, 1] -> normalised ``[bs, 3, S / f, S / f]``.  As in the reference (:152-156) every image that is at
        least 512 wide goes through ``self.downsample``, whose factor ``f = self.size // 512`` is fixed by the constructor and NOT derived
        from the image: the default ``size=1024`` parser maps 1024 -> 512 (the only case on the swap path), 512 -> 256, 2048 -> 1024.
        ``downsample=False`` is the reference's other branch (:157-160, images narrower than 512 after their PIL resize to 512): clamp
        and normalise only."""
        f = self.downsample.factor if downsample else 1
        if f == 1:
            if pm1:
                img01 = (img01 + 1) / 2
            return ops.bicubic_down_normalize(img01, None, 1, self._mean, self._std, out=out)     # clamp + normalise only
        if img01.shape[-1] % f or img01.shape[-2] % f:
            raise ValueError(f"image size {tuple(img01.shape[-2:])} is not a multiple of the parser's down-sampling factor {f}")
        return ops.bicubic_down_normalize(img01, self.downsample.taps.to(img01.device), f, self._mean, self._std, out=out, pm1=pm1)

    def parse_batch(self, img01, seg12=True, pm1=False):
        """``[bs, 3, S, S]`` in [0, 1] -> uint8 labels ``[bs, 512, 512]`` (12-class when ``seg12``).  ``pm1``: the images are in [-1, 1] (the
        pipeline's range) and ``(img + 1) / 2`` happens inside the down-sampling kernel.  ``img01`` may be a sequence of such batches (same
        size each): they are parsed as ONE batch — every part is down-sampled into its slice of the network's input, no concatenated copy of
        the full-size images is made."""
        with torch.no_grad():
            if isinstance(img01, (list, tuple)):
                f = self.downsample.factor
                n = [int(t.shape[0]) for t in img01]
                first = img01[0]
                x = torch.empty((sum(n), first.shape[1], first.shape[2] // f, first.shape[3] // f), dtype=torch.float32, device=first.device)
                lo = 0
                for t, k in zip(img01, n):
                    if tuple(t.shape[1:]) != tuple(first.shape[1:]):
                        raise ValueError("parse_batch: the parts of a batch must have one image size")
                    self.preprocess_tensor(t, pm1=pm1, out=x[lo:lo + k])
                    lo += k
            else:
                x = self.preprocess_tensor(img01, pm1=pm1)
            return self.seg.parse(x, self._lut12 if seg12 else None)

    # ---- reference API -------------------------------------------------------------------------------------------
    def preprocess_img(self, img):
        if img.size[0] >= 512:
            return self.preprocess_tensor(_pil_to_tensor01(img, self.device))
        im = img.resize((512, 512), Image.BILINEAR)                                    # reference :157-159
        return self.preprocess_tensor(_pil_to_tensor01(im, self.device), downsample=False)

    def forward(self, img):
        """PIL image -> LongTensor ``[512, 512]`` of 19-class labels (reference :162-176)."""
        im = self.preprocess_img(img)
        return self.seg.parse(im)[0].long()


def init_faceParsing_pretrained_model(ckpt_path):
    parser = FaceParser(seg_ckpt=ckpt_path)
    print("Load faceParsing pre-traiend model success!")
    return parser


def faceParsing_demo(model, img, convert_to_seg12=True):
    """reference :187-200: uint8 numpy ``[512, 512]`` label map of a PIL image (12-class when ``convert_to_seg12``)."""
    with torch.no_grad():
        im = model.preprocess_img(img)
        seg = model.seg.parse(im, model._lut12 if convert_to_seg12 else None)[0]
    return seg.cpu().numpy().astype(np.uint8)


def vis_parsing_maps(image, parsing_anno, stride=1):
    """Overlay of the label map on the image (reference :87-129), BGR uint8 like the cv2 original; numpy only."""
    part_colors = [[255, 0, 0], [255, 85, 0], [255, 170, 0], [255, 0, 85], [255, 0, 170], [0, 255, 0], [85, 255, 0], [170, 255, 0],
                   [0, 255, 85], [0, 255, 170], [0, 0, 255], [85, 0, 255], [170, 0, 255], [0, 85, 255], [0, 170, 255], [255, 255, 0],
                   [255, 255, 85], [255, 255, 170], [255, 0, 255], [255, 85, 255], [255, 170, 255], [0, 255, 255], [85, 255, 255],
                   [170, 255, 255]]
    im = np.array(image.resize((parsing_anno.shape[0], parsing_anno.shape[1]), Image.BILINEAR)).astype(np.uint8)
    anno = parsing_anno.copy().astype(np.uint8)
    if stride != 1:
        anno = np.repeat(np.repeat(anno, int(stride), axis=0), int(stride), axis=1)
    color = np.zeros((anno.shape[0], anno.shape[1], 3)) + 255
    for pi in range(1, int(np.max(anno)) + 1):
        color[anno == pi] = part_colors[pi]
    bgr = im[:, :, ::-1].astype(np.float64)
    return np.clip(np.rint(0.4 * bgr + 0.6 * color), 0, 255).astype(np.uint8)
