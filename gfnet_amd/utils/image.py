"""Image loading / resize / normalisation in front of the hot path (reference: model/network.py:
287-320 with utils/utils.py:18-27, 84-113).  Plain torch -- this is host-side plumbing around the
backbone, not part of the accelerated path; torchvision is not required.
"""
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def to_tensor(im):
    """PIL image -> float CHW tensor in [0,1] (torchvision ToTensor / ToTensorScaled)."""
    if isinstance(im, torch.Tensor):
        return im
    a = np.asarray(im.convert("RGB"), dtype=np.float32).transpose(2, 0, 1) / 255.0
    return torch.from_numpy(np.ascontiguousarray(a))


def load_pair(im0, im1, batched=True):
    """The three input kinds GFNet.match accepts (network.py:287,299,311): path, PIL image, tensor.
    Returns two (1,3,H,W) tensors and whether the result should stay batched."""
    from PIL import Image

    if isinstance(im0, (str, Path)):
        im0, im1 = Image.open(im0).convert("RGB"), Image.open(im1).convert("RGB")
    elif isinstance(im0, Image.Image):
        batched = False
    elif isinstance(im0, torch.Tensor):
        batched = False
    else:
        raise TypeError(f"unsupported image type {type(im0)}")
    a, b = to_tensor(im0), to_tensor(im1)
    a = a[None] if a.dim() == 3 else a
    b = b[None] if b.dim() == 3 else b
    return a, b, batched


def resize_normalise(im, size):
    """TupleResize(size, BICUBIC, antialias) + ImageNet TupleNormalize on a (1,3,H,W) tensor."""
    h, w = size
    x = im[:, :3].float()
    if tuple(x.shape[-2:]) != (h, w):
        x = F.interpolate(x, size=(h, w), mode="bicubic", align_corners=False, antialias=True)
    mean = torch.tensor(IMAGENET_MEAN, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=x.device).view(1, 3, 1, 1)
    return (x - mean) / std
