"""Drop-in for the reference's utils/kde.py (same name, same signature)."""
import torch

from .. import ops


def kde(x, std=0.1, half=True, down=None):
    """Gaussian kernel density of the rows of x (N,D) against x[::down] (utils/kde.py:4-13).

    The N x M score matrix of the reference is never formed (csrc/kde.hip streams it).  With
    half=True the inputs are rounded to fp16 like the reference does and the result is returned as
    fp16, but the distances and the sum are still fp32 -- the reference's fp16 cdist is 12 % off the
    exact density, this is not reproduced.
    """
    if half:
        x = x.half()
    xf = x.float().contiguous()
    N, D = xf.shape
    if down is None or int(down) == 1:
        dens = ops.kde_density(xf, None, std=std)
    else:
        dens = ops.kde_density(xf, xf, std=std, y_row_stride=int(down) * D)
    return dens.half() if half else dens
