"""Image loading / resize / normalisation in front of the backbone (reference: model/network.py:
287-346 with utils/utils.py:18-27, 84-116).  Decoding stays PIL/numpy on the host; resize +
normalise is one HIP kernel (csrc/grid_ops.hip).  torchvision is not required.
"""
from pathlib import Path

import numpy as np
import torch

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
    Returns two (1,3,H,W) tensors in [0,1], whether the result stays batched, and the resize mode the reference
    uses for the first pass of that kind (it passes mode=2 = bilinear for paths, the bicubic default otherwise)."""
    from PIL import Image

    mode = "bicubic"
    if isinstance(im0, (str, Path)):
        im0, im1 = Image.open(im0).convert("RGB"), Image.open(im1).convert("RGB")
        mode = "bilinear"
    elif isinstance(im0, Image.Image):
        batched = False
    elif isinstance(im0, torch.Tensor):
        batched = False
    else:
        raise TypeError(f"unsupported image type {type(im0)}")
    a, b = to_tensor(im0), to_tensor(im1)
    a = a[None] if a.dim() == 3 else a
    b = b[None] if b.dim() == 3 else b
    return a, b, batched, mode


def resize_normalise(im, size, mode="bicubic"):
    """get_tuple_transform_ops(resize=size, mode, normalize=True) (utils/utils.py:18-27) on a (B,3+,H,W) tensor in
    [0,1]: torchvision's Resize on a float tensor with antialias=None is F.interpolate(mode, align_corners=False) without
    antialiasing; then ImageNet Normalize.  GPU tensors go through the HIP kernel (ops.resize_normalise); CPU tensors are
    moved to the GPU first (this package has no CPU arithmetic)."""
    from .. import ops

    if not im.is_cuda:
        im = im.cuda()
    return ops.resize_normalise(im, size, mode)
