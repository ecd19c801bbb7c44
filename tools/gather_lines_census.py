"""CPU census: how many cache lines a 64-lane x_hat pair gather of refiner_input touches under the bench's homography flows, by the
shape of a wave's 64 cells (profiles/r06_refiner_input_lines.md).  python tools/gather_lines_census.py [size G]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfnet_amd._synthetic import random_homographies, warp_grid  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 448
G = int(sys.argv[2]) if len(sys.argv) > 2 else 256
W = size
gen = torch.Generator().manual_seed(0)
H = random_homographies(8, size, gen)
Hs = np.concatenate([H, np.linalg.inv(H)])
Hs = Hs / Hs[:, 2:3, 2:3]
g = warp_grid(Hs, G, size, "cpu").numpy()
x = ((g[..., 0] + 1) * W - 1) / 2
y = ((g[..., 1] + 1) * W - 1) / 2
x0 = np.clip(np.floor(x).astype(int), 0, W - 2)
y0 = np.clip(np.floor(y).astype(int), 0, W - 1)
addr = (y0 * W + x0) * 4  # byte address of a lane's pixel pair inside a channel plane


def lines(shape, LB):
    bh, bw = shape
    a = addr.reshape(-1, G // bh, bh, G // bw, bw).transpose(0, 1, 3, 2, 4).reshape(-1, bh * bw)
    l0, l1 = a // LB, (a + 7) // LB
    return np.mean([len(set(r0.tolist()) | set(r1.tolist())) for r0, r1 in zip(l0[::7], l1[::7])])


for shape in [(1, 64), (2, 32), (4, 16), (8, 8)]:
    print(f"wave = {shape[0]} x {shape[1]} cells: {lines(shape, 128):5.1f} 128-byte lines, {lines(shape, 64):5.1f} 64-byte sectors per x_hat gather")
cx = (np.arange(G) + 0.5) * W / G - 0.5
print("regular grid_feature gather of a 1 x 64 wave:", len(set((np.floor(cx[:64]).astype(int) * 4 // 128).tolist())), "lines")
