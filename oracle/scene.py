"""A synthetic scene (gfnet_amd/_synthetic.py) walked through the oracle on the host cores -- the checker of the whole-path
parity tests and the `cpu_baseline` leg of bench.py.  Test infrastructure like the rest of oracle/: never imported by gfnet_amd/.
"""
import math

import numpy as np
import torch

import oracle

SCALES = ["16", "8", "4", "2", "1"]


def cpu_pair(scene, b, np_pyr, np_up, np_gt, np_noise, seed, return_all=False):
    """Pair b of `scene` (gfnet_amd._synthetic.Scene) through the oracle: both passes of the coarse-to-fine loop
    (model/network.py:230-281, 326-349), match_post, sample, convert_coordinates and the homography solve."""
    self = scene
    m, nb = self.model, self.B

    def run_pass(p0, p1, size, grids, radii, itrs, scl, pre=None, sf=1.0):
        f0 = {s: np.concatenate((p0[s][b:b + 1], p1[s][b:b + 1])) for s in scl}
        f1 = {s: np.concatenate((p1[s][b:b + 1], p0[s][b:b + 1])) for s in scl}
        res = {}
        for i, s in enumerate(scl):
            if i == 0:
                if pre is None:
                    flow = oracle.corr_softargmax(f0[s], f1[s])
                    cert = np.zeros((2, 1) + flow.shape[2:], np.float32)
                else:
                    flow = oracle.interpolate_bilinear(pre[0], grids[0])
                    cert = oracle.interpolate_bilinear(pre[1], grids[0])
            ref = m.conv_refiner[s].inner
            G = grids[i]
            disp_prev = np.full_like(flow, 1e-7)
            for itr in range(itrs[i]):
                oracle.refiner_input(G, f0[s], f1[s], flow, ref.disp_emb.weight.detach().cpu().numpy(),
                                     ref.disp_emb.bias.detach().cpu().numpy(), radii[i], scale_factor=sf,
                                     corr_in_other=radii[i] > 0)
                target = np_gt[G][[b, b + nb]] + np_noise[G][itr][[b, b + nb]]
                dl = (target - flow) * np.float32(4.0 * size / int(s))
                flow, cert, disp_prev, rel = oracle.flow_update(flow, cert, dl, np.ones_like(cert), disp_prev, int(s), size, size,
                                                                return_rel=True)
                res[(s, itr + 1)] = (flow, cert, rel)
            res[s] = (flow, cert)
            if s != "1":
                flow = oracle.interpolate_bilinear(flow, grids[i + 1])
                cert = oracle.interpolate_bilinear(cert, grids[i + 1])
        return res

    r1 = run_pass(np_pyr[0], np_pyr[1], self.size, self.grids, m.radius, self.num_itr, SCALES)
    gu, ru, iu = m.upsample_grids(self.up)
    r2 = run_pass(np_up[0], np_up[1], self.up, gu, ru, iu, SCALES[1:], pre=r1["1"], sf=math.sqrt(self.up * self.up / (self.size * self.size)))
    warp, cert = oracle.match_post(r2["1"][0], r2["1"][1], r1["16"][1], symmetric=True, attenuate_cert=True)
    if return_all:
        return r1, r2, warp, cert
    torch.manual_seed(1234 + b)
    good, _ = oracle.sample(warp[0], cert[0], num=5000, device_is_gpu=True)
    pts = oracle.convert_matches(good, *self.sizes)
    return oracle.homography_ransac(pts[None], thresh=3.0, iters=2000, seed=seed)
