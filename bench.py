#!/usr/bin/env python3
"""bench.py -- pairs/sec of the GFNet post-backbone hot path on MI355X (BASELINE.json metric).

One "step" = one batch of synthetic image pairs through the whole path on feature pyramids that are
already resident in HBM:
  global correlation + soft-argmax (scale 16) -> per scale and refiner iteration: refiner input (grid
  features, x_hat, displacement embedding, local correlation into the concat buffer) + flow update +
  inter-scale resize, for the first pass and the 1.25x refinement pass (test.py defaults: symmetric,
  upsample_preds, attenuate_cert) -> match post-processing -> balanced sampling (2 draws without
  replacement + KDE, N=M=20000) -> device RANSAC/DLT/LM homography solve -> (N>1) RCCL all-gather of H.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 448b32|672b16|pyr-fp16] [--conv-stack off|fp32|fp16|amp]

Workloads (BASELINE.json configs):
  448b32   (default, configs[1]) 448x448, 32 pairs per GPU, basic.json (num_itr 1), fp32 features
  672b16   (configs[2]) googlemap 672x672, 16 pairs per GPU, map.json (num_itr 2 per scale), grids by the rule of
           network.py:329 ([S/14, S/14, 2S/14, 4S/14, 8S/14] = 48/48/96/192/384, refinement pass at 840)
  pyr-fp16 (configs[4]) feature pyramids stored in fp16 at the 224 / 448 / 672 test-set sizes, 8 pairs each per step
The refiners' conv stacks are a stand-in by default (`config.excluded`); `--conv-stack fp32|fp16|amp`
runs the reference's architecture on the HIP conv-stack kernels inside the timed region (fp32: fp32 throughout; fp16: fp16
operands of the 1x1 convs; amp: fp16 operands and fp16 maps between the blocks -- the class the reference's amp=True refiners
run in, model/network.py:560-562).  A default single-GPU run also reports that last figure as `with_conv_stacks` next to
`value` (secondary measurement after the timed region; --no-stack-leg skips it).

Prints ONE JSON line on rank 0 (driver contract).  `roofline` is for the dominant kernel of the named config, the scale-4
local-correlation call of the first pass (c32, r=4), timed with HIP events on its launch stream inside the timed steps;
`cpu_baseline` is the C/OpenMP oracle (a port of the reference's algorithm, pinned on reference-generated goldens) on a
bounded sample of the same workload: 1 warm-up + median of 3, rank 0, N=1 only.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

FEAT = {"16": 64, "8": 64, "4": 32, "2": 16, "1": 8}
DISP = {"16": 64, "8": 64, "4": 32, "2": 16, "1": 8}
RADIUS = [7, 6, 4, 2, 0]
SCALES = ["16", "8", "4", "2", "1"]
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
FLOW_NOISE_PX = 0.5    # true flow + N(0, (0.5/S)^2) in normalised units

WORKLOADS = {
    "448b32": {"sizes": [448], "pairs": 32, "num_itr": [1] * 5, "dtype": "fp32", "cpu_pairs": 2,
               "label": "448x448 batch=32 synthetic pairs per GPU, local_correlation radius 7/6/4/2 (BASELINE configs[1])"},
    "672b16": {"sizes": [672], "pairs": 16, "num_itr": [2] * 5, "dtype": "fp32", "cpu_pairs": 1,
               "label": "googlemap 672x672 batch=16 per GPU, num_itr=[2]*5 (gfnet_configs/map.json), grids 48/48/96/192/384 "
                        "(BASELINE configs[2])"},
    "pyr-fp16": {"sizes": [224, 448, 672], "pairs": 8, "num_itr": [1] * 5, "dtype": "fp16", "cpu_pairs": 1,
                 "label": "multi-scale 224/448/672 pyramids stored in fp16, 8 pairs per size and step, streamed KDE + device solve "
                          "(BASELINE configs[4])"},
}


def grids_for(size):
    """num_grid of a pass at image size `size`: network.py:329's rule, [hs/14, 2x, 4x, 8x] with the coarsest repeated for
    scale 16 (basic.json's [32,32,64,128,256] at 448)."""
    g = int(size / 14)
    return [g, g, 2 * g, 4 * g, 8 * g]


def side_of(scale, size):
    return size // 14 if scale == "16" else size // int(scale)  # network.py:185-198: 32/56/112/224/448 at 448


def random_homographies(n, size, gen):
    """4-corner perturbation U(-0.15 S, 0.15 S), as (n,3,3) float64."""
    src = np.array([[0, 0], [size - 1, 0], [size - 1, size - 1], [0, size - 1]], np.float64)
    out = []
    for _ in range(n):
        dst = src + (torch.rand(4, 2, generator=gen, dtype=torch.float64).numpy() * 2 - 1) * 0.15 * size
        A = []
        for (x, y), (u, v) in zip(src, dst):
            A.append([x, y, 1, 0, 0, 0, -u * x, -u * y, -u])
            A.append([0, 0, 0, x, y, 1, -v * x, -v * y, -v])
        h = np.linalg.svd(np.array(A))[2][-1].reshape(3, 3)
        out.append(h / h[2, 2])
    return np.stack(out)


def warp_grid(H, side, size, device):
    """Normalised sampling grid (n,side,side,2): where each cell centre of a side x side map of the
    A image lands in the B image under H (pixel coordinates of a size x size image)."""
    lin = (torch.arange(side, dtype=torch.float64) * 2 + 1) / side - 1
    gy, gx = torch.meshgrid(lin, lin, indexing="ij")
    px, py = (size - 1) * (gx + 1) / 2, (size - 1) * (gy + 1) / 2
    Ht = torch.from_numpy(H)
    w = Ht[:, 2, 0, None, None] * px + Ht[:, 2, 1, None, None] * py + Ht[:, 2, 2, None, None]
    u = (Ht[:, 0, 0, None, None] * px + Ht[:, 0, 1, None, None] * py + Ht[:, 0, 2, None, None]) / w
    v = (Ht[:, 1, 0, None, None] * px + Ht[:, 1, 1, None, None] * py + Ht[:, 1, 2, None, None]) / w
    return torch.stack((2 * u / (size - 1) - 1, 2 * v / (size - 1) - 1), -1).float().to(device)


def make_pyramids(H, size, scales, device, gen, dtype):
    """Synthetic feature pyramids: B-image features = smoothed noise (amplitude 2), A-image features =
    the B features seen through H + 0.1 noise, so correlation peaks and flows are meaningful."""
    n = H.shape[0]
    pa, pb = {}, {}
    for s in scales:
        side, c = side_of(s, size), FEAT[s]
        fb = F.avg_pool2d(torch.randn(n, c, side, side, device=device, generator=gen), 3, 1, 1) * 6.0
        fa = F.grid_sample(fb, warp_grid(H, side, size, device), mode="bilinear", padding_mode="zeros", align_corners=False)
        fa = fa + 0.1 * torch.randn(n, c, side, side, device=device, generator=gen)
        pa[s], pb[s] = fa.to(dtype).contiguous(), fb.to(dtype).contiguous()
    return pa, pb


class StandInRefiner(nn.Module):
    """The HIP part of ConvRefiner.forward (network.py:533-558) followed by a stand-in for the conv
    stack (network.py:560-563): like a trained refiner it returns the increment that moves the flow onto the
    true warp plus sub-pixel noise (N(0, (0.5/S)^2), a fresh seeded realisation per refiner iteration) and a
    constant certainty increment.  One torch elementwise op; everything else is the real path.  With
    --conv-stack the reference's conv stack runs too (random-init, its output weighted 0)."""

    def __init__(self, feat, disp, radius, scale, targets, num_itr, conv_stack="off"):
        super().__init__()
        from gfnet_amd.model.network import ConvRefiner, _refiner_for

        K = (2 * radius + 1) ** 2 if radius > 0 else 0
        dim = 2 * feat + disp + K
        if conv_stack == "off":
            self.inner = ConvRefiner(dim, dim, 3, kernel_size=5, dw=True, hidden_blocks=0, displacement_emb="linear",
                                     displacement_emb_dim=disp, local_corr_num=radius, corr_in_other=radius > 0)
        else:
            self.inner = _refiner_for(feat, disp, radius)
            self.inner.conv_precision = conv_stack
        self.conv_stack = conv_stack
        self.scale, self.num_itr = scale, num_itr
        self.targets = targets  # {num_grid: ([k * (true flow + noise_itr) for itr], k)}
        self._cert, self._calls = {}, {}

    supports_reuse_d = True  # GFNet.forward_pyramids: later iterations at a scale keep the grid_feature planes

    @property
    def last_d(self):
        return self.inner.last_d

    @last_d.setter
    def last_d(self, v):
        self.inner.last_d = v

    def forward(self, num_grid, x, y, flow, scale_factor=1, reuse_d=None):
        d, lc = self.inner.assemble(num_grid, x, y, flow, scale_factor, reuse=reuse_d)
        tk, k = self.targets[num_grid]
        itr = self._calls.get(num_grid, 0)
        self._calls[num_grid] = (itr + 1) % self.num_itr
        delta = torch.add(tk[itr], flow, alpha=-k)  # (gt + noise - flow) * k in one launch; k undone by network.py:262-263
        if num_grid not in self._cert:
            self._cert[num_grid] = torch.full((flow.shape[0], 1, num_grid, num_grid), 1.0, device=flow.device)
        cert = self._cert[num_grid]
        if self.conv_stack != "off":
            out = self.inner.conv_stack(d)
            delta = torch.addcmul(delta, out[:, :2], torch.zeros((), device=d.device))
            cert = torch.addcmul(cert, out[:, 2:3], torch.zeros((), device=d.device))
        return delta, cert, lc


class Scene:
    """Everything one image size needs: pyramids of both passes, true warps + noise on every grid, the model."""

    def __init__(self, size, pairs, num_itr, dtype, conv_stack, dev, rank, upsample=True):
        from gfnet_amd.model.network import GFNet

        self.size, self.up, self.B, self.num_itr = size, int(size * 1.25), pairs, num_itr
        gen_cpu = torch.Generator().manual_seed(1000 + rank + 7 * size)
        gen = torch.Generator(device=dev).manual_seed(2000 + rank + 7 * size)
        S0, S1 = self.size, self.up
        self.H = random_homographies(pairs, S0, gen_cpu)
        self.pyr = make_pyramids(self.H, S0, SCALES, dev, gen, dtype)
        Hup = np.stack([np.diag([S1 / S0, S1 / S0, 1.0]) @ h @ np.diag([S0 / S1, S0 / S1, 1.0]) for h in self.H])
        self.pyr_up = make_pyramids(Hup, S1, SCALES[1:], dev, gen, dtype) if upsample else (None, None)
        self.grids, self.grids_up = grids_for(S0), grids_for(S1)[1:]
        # true normalised warps on every grid the two passes use (A->B for the first B rows, B->A after), plus the
        # per-iteration noise realisations (generated on the CPU so that the oracle leg sees the same bits)
        self.gt, self.noise = {}, {}
        Hinv, Hupinv = np.linalg.inv(self.H), np.linalg.inv(Hup)
        passes = [(self.grids, self.H, Hinv, S0)] + ([(self.grids_up, Hup, Hupinv, S1)] if upsample else [])
        for grids, Hf, Hb, S in passes:
            for G in set(grids):
                self.gt[G] = torch.cat((warp_grid(Hf, G, S, dev), warp_grid(Hb, G, S, dev))).permute(0, 3, 1, 2).contiguous()
                self.noise[G] = [torch.randn(2 * pairs, 2, G, G, generator=gen_cpu) * (FLOW_NOISE_PX / S) for _ in range(max(num_itr))]
        targets = {s: {} for s in SCALES}  # per scale: {num_grid: ([k * (true flow + noise) per iteration], k)}
        for i, s in enumerate(SCALES):
            uses = [(self.grids[i], S0)] + ([(self.grids_up[i - 1], S1)] if upsample and i >= 1 else [])
            for G, S in uses:
                k = 4.0 * S / int(s)  # undone by network.py:262-263's scale / (4 * W0)
                targets[s][G] = ([(self.gt[G] + n.to(dev)) * k for n in self.noise[G][:num_itr[i]]], k)
        refiners = nn.ModuleDict({s: StandInRefiner(FEAT[s], DISP[s], RADIUS[i], int(s), targets[s], num_itr[i], conv_stack)
                                  for i, s in enumerate(SCALES)})
        conf = {"encoder_cfg": {"feat_chs": [64, 32, 16, 8]},
                "matcher": {"num_grid": self.grids, "radius": RADIUS, "displacement_dim": [64, 64, 32, 16, 8], "num_itr": num_itr}}
        self.model = GFNet(conf, initial_res=(S0, S0), upsample_res=(S1, S1), symmetric=True, upsample_preds=upsample,
                           attenuate_cert=True, conv_refiner=refiners).to(dev).eval()
        self.sizes = (S0, S0, S0, S0)
        self.roofline_key = f"local_corr_c32_h{side_of('4', S0)}_g{self.grids[2]}_r4"

    def step(self, seed):
        from gfnet_amd.estimation import estimate_homographies
        from gfnet_amd.model.network import sample_batched

        m = self.model
        warp, cert = m.match_pyramids(self.pyr[0], self.pyr[1], self.pyr_up[0], self.pyr_up[1], batched=True)
        good, _ = sample_batched(m, warp, cert, 5000)
        Hl = estimate_homographies(good, self.sizes, iters=m.ransac_iters, seed=seed)
        return Hl, good

    # ---- the same stages through the C/OpenMP oracle on pair b (host cores) --------------------------------
    def cpu_pair(self, b, np_pyr, np_up, np_gt, np_noise, seed, return_all=False):
        import oracle

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


def algorithmic_bytes_local_corr(B, c, hs, G, r, feat_bytes=4):
    """f0 + f1 + flow + out: 3 489 792 B per pair-direction at scale 4 of the 448 pass in fp32.
    f0 is the grid_feature slice of the concat buffer (always fp32 there); f1 is stored in `feat_bytes`."""
    return B * (4 * c * G * G + feat_bytes * c * hs * hs + 8 * G * G + 4 * (2 * r + 1) ** 2 * G * G)


def cpu_baseline(scenes, n_pairs):
    """1 warm-up + median of 3 passes of `n_pairs` pairs of every scene through oracle/ (host cores)."""
    import oracle

    to_np = lambda p: {s: t[:n_pairs].float().cpu().numpy() for s, t in p.items()}  # noqa: E731
    prepared = []
    for sc in scenes:
        np_gt = {G: t.cpu().numpy() for G, t in sc.gt.items()}
        np_noise = {G: [n.numpy() for n in ns] for G, ns in sc.noise.items()}
        prepared.append((sc, (to_np(sc.pyr[0]), to_np(sc.pyr[1])), (to_np(sc.pyr_up[0]), to_np(sc.pyr_up[1])), np_gt, np_noise))
    times = []
    for rep in range(4):
        t0 = time.time()
        for sc, npyr, nup, ngt, nnoise in prepared:
            for b in range(n_pairs):
                sc.cpu_pair(b, npyr, nup, ngt, nnoise, seed=b)
        times.append(time.time() - t0)
    dt = float(np.median(times[1:]))
    return n_pairs * len(scenes) / dt, oracle.max_threads()


def solve_parity(scene, good, Hl, n):
    """Corner error (px) between the device H and the oracle H on identical (device-sampled) matches, and of the device H
    against the ground-truth H."""
    import oracle

    pts = oracle.convert_matches(good[:n].cpu().numpy(), *scene.sizes)
    Ho, _, _ = oracle.homography_ransac(pts, thresh=3.0, iters=2000, seed=0)
    Hd = Hl[:n].cpu().numpy()
    S = scene.size
    vs_oracle = [oracle.corner_error(Ho[i], Hd[i], S, S, clamp=1e9) for i in range(n)]
    vs_truth = [oracle.corner_error(scene.H[i], Hd[i], S, S, clamp=1e9) for i in range(n)]
    return float(np.mean(vs_oracle)), float(np.mean(vs_truth))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=tuple(WORKLOADS), default="448b32")
    ap.add_argument("--pairs-per-gpu", type=int, default=0, help="override the workload's pairs per GPU (per size)")
    ap.add_argument("--cpu-pairs", type=int, default=-1, help="pairs per size for the CPU-oracle baseline leg (0 = skip; default: per workload)")
    ap.add_argument("--breakdown", action="store_true", help="print per-stage GPU times of one step to stderr")
    ap.add_argument("--no-stack-leg", action="store_true",
                    help="skip the secondary measurement `with_conv_stacks` (default single-GPU runs with --conv-stack off also time the same "
                         "step with the refiners' conv stacks in the reference's autocast class)")
    ap.add_argument("--conv-stack", choices=("off", "fp32", "fp16", "amp"), default="off",
                    help="also run the refiners' conv stacks (reference architecture, random-init) on the HIP kernels, with "
                         "fp32 or fp16 1x1-conv operands; default off = the north-star hot path only")
    args = ap.parse_args()

    from gfnet_amd import ops, parallel

    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the hot path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    wl = WORKLOADS[args.workload]
    B = args.pairs_per_gpu or wl["pairs"]
    dtype = torch.float16 if wl["dtype"] == "fp16" else torch.float32

    # ---- synthetic workload, resident in HBM before the timed region -------------------------------
    scenes = [Scene(S, B, wl["num_itr"], dtype, args.conv_stack, dev, rank) for S in wl["sizes"]]
    main_scene = scenes[min(1, len(scenes) - 1)] if len(scenes) > 1 else scenes[0]  # the 448 scene of the pyramid workload
    pairs_per_step = B * len(scenes)

    def step(seed):
        outs = [sc.step(seed) for sc in scenes]
        Hl = torch.cat([o[0] for o in outs])
        return parallel.gather_homographies(Hl), outs

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.inference_mode():
        for i in range(args.warmup):
            step(i)
        sync()
        if args.breakdown and rank == 0:
            from gfnet_amd.estimation import estimate_homographies
            from gfnet_amd.model.network import sample_batched

            def timed(fn):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
                return r, e0.elapsed_time(e1)
            for sc in scenes:
                m = sc.model
                (warp, cert), t_match = timed(lambda: m.match_pyramids(sc.pyr[0], sc.pyr[1], sc.pyr_up[0], sc.pyr_up[1], batched=True))
                (gm, _), t_sample = timed(lambda: sample_batched(m, warp, cert, 5000))
                _, t_solve = timed(lambda: estimate_homographies(gm, sc.sizes, iters=m.ransac_iters, seed=0))
                print(f"[breakdown] {sc.size}: match({sc.size}+{sc.up}) {t_match:.2f} ms | sample {t_sample:.2f} ms | solve {t_solve:.2f} ms", file=sys.stderr)
        ops.kernel_events = {main_scene.roofline_key: []}
        t0 = time.perf_counter()
        for i in range(args.steps):
            Hall, outs = step(0)
        sync()
        dt = time.perf_counter() - t0
        events = ops.kernel_events[main_scene.roofline_key]
        ops.kernel_events = None
        # one more, untimed, step with a device sync after every local-correlation call: how many tiles the second launch
        # took and how many cells were redone per tap (both depend on the flows the workload produces)
        ops.kernel_counters = {}
        step(0)
        counters = ops.kernel_counters.get(main_scene.roofline_key, [])
        if args.breakdown and rank == 0:
            for name, cs in ops.kernel_counters.items():
                print(f"[counters] {name}: second-launch tiles / flagged cells / half-staged tiles per call: {cs}", file=sys.stderr)
        ops.kernel_counters = None
        # secondary figure: the same step with the refiners' real conv stacks (reference architecture, random-init)
        # in the class the reference runs them in on a GPU -- `value` above replaces them by a one-op stand-in
        stack_leg = None
        if args.conv_stack == "off" and world == 1 and not args.no_stack_leg:
            with torch.inference_mode(False):  # module parameters must be ordinary tensors (the packed-parameter cache reads their versions)
                scenes2 = [Scene(S, B, wl["num_itr"], dtype, "amp", dev, rank) for S in wl["sizes"]]
            for i in range(2):
                for sc in scenes2:
                    sc.step(i)
            torch.cuda.synchronize()
            n2 = max(3, min(args.steps, 10))
            t1 = time.perf_counter()
            for i in range(n2):
                for sc in scenes2:
                    sc.step(0)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            stack_leg = {"value": round(pairs_per_step * n2 / dt2, 2), "unit": "pairs/s", "ms_per_step": round(dt2 / n2 * 1e3, 3), "steps": n2,
                         "refiner_conv_stack": "reference architecture, random-init, HIP conv_stack kernels, conv_precision='amp' (fp16 maps, "
                                               "fp16 operands, fp32 accumulation: model/network.py:560-562)"}
            del scenes2
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    pairs_per_s = world * pairs_per_step * args.steps / dt

    S0 = main_scene.size
    hs4, G4 = side_of("4", S0), main_scene.grids[2]
    n_calls = wl["num_itr"][2]  # roofline op calls per step and scene (one per refiner iteration at scale 4)
    kern_us = float(np.mean([a.elapsed_time(b) for a, b in events])) * 1e3 if events else float("nan")
    fbytes = 2 if dtype == torch.float16 and ops.NATIVE_FP16 else 4
    nbytes = algorithmic_bytes_local_corr(2 * B, 32, hs4, G4, 4, fbytes)
    achieved = nbytes / (kern_us * 1e-6) / 1e9 if events else float("nan")
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", "local_corr_pmc.json")  # written from the rocprofv3 --pmc passes (tools/pmc_hbm_local_corr.sh)
    if os.path.exists(pmc) and args.workload == "448b32":
        try:
            j = json.load(open(pmc))
            traffic, traffic_src = j.get("hbm_bytes_per_launch"), "static: " + j.get("source", "profiles/local_corr_pmc.json")
        except Exception:
            traffic = None
    tiles4 = 2 * B * ((G4 + 15) // 16) * ((G4 + 3) // 4)
    irregular = float(np.mean([c[0] for c in counters])) / tiles4 if counters else None
    flagged = float(np.mean([c[1] for c in counters])) / (2 * B * G4 * G4) if counters else None
    halves = float(np.mean([c[2] for c in counters])) / tiles4 if counters else None

    out = {
        "metric": "image pairs/sec at 448x448 (post-backbone hot path: correlation -> flow -> sampling -> homography)"
                  if args.workload == "448b32" else f"image pairs/sec, workload {args.workload} (post-backbone hot path)",
        "value": round(pairs_per_s, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl["label"], "workload_key": args.workload, "pairs_per_gpu": pairs_per_step,
                   "image_sizes": wl["sizes"], "num_itr": wl["num_itr"], "feature_storage": wl["dtype"],
                   "symmetric": True, "upsample_pass": "1.25x (560 at 448)", "attenuate_cert": True,
                   "flow_noise": f"stand-in increment = true warp + N(0,({FLOW_NOISE_PX}/S)^2) - flow, fresh realisation per iteration",
                   "stages": "corr_softargmax, (refiner_input + local_corr + flow_update) x scales x num_itr for both passes, resize, "
                             "match_post, sample(2 draws without replacement + KDE 20000^2), RANSAC(<= 2000 hypotheses, OpenCV's confidence-0.99999 "
                             "bound)+DLT+LM, H all-gather",
                   "excluded": ("DINOv2/FPN backbone; refiner conv stacks replaced by the stand-in increment (1 torch elementwise op per "
                                "refiner call; --conv-stack fp32|fp16|amp runs them on the HIP conv-stack kernels)") if args.conv_stack == "off"
                   else "DINOv2/FPN backbone (PyTorch-ROCm host code)",
                   "refiner_conv_stack": "off" if args.conv_stack == "off" else
                   f"reference architecture (9 dw5x5+BN+ReLU+1x1 blocks + out conv per refiner call, C=417/361/177/73/24), random-init, "
                   f"HIP conv_stack kernels, " + {"fp32": "fp32 throughout", "fp16": "1x1 operands fp16, maps fp32",
                                               "amp": "fp16 maps between the blocks, depthwise and 1x1 operands fp16, fp32 accumulation "
                                                      "(the reference's autocast class)"}[args.conv_stack] +
                   "; output weighted 0 next to the stand-in increment",
                   "parallelism": f"pairs sharded over {world} GPU(s), RCCL all-gather of H only"},
        "roofline": {"bound": "hbm",
                     "kernel": f"gfn_local_corr_fwd_dt call (lean tile kernel, its first workgroups finish the tiles the plan left to "
                               f"the second launch; the plan itself is written by the refiner_input launch; c32, {hs4}x{hs4}, G{G4}, r4, "
                               f"{2 * B} directions)",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": round(kern_us, 2), "calls_per_step": n_calls,
                     "irregular_tile_frac": irregular, "half_staged_tile_frac": halves, "flagged_cell_frac": flagged},
    }
    n_cpu = wl["cpu_pairs"] if args.cpu_pairs < 0 else args.cpu_pairs
    if rank == 0 and world == 1 and n_cpu > 0:
        n_cpu = min(n_cpu, B)
        v, cores = cpu_baseline(scenes, n_cpu)
        out["cpu_baseline"] = {"value": round(v, 4), "unit": "pairs/s", "cores": cores, "kind": "port",
                               "sample": f"{n_cpu} pair(s) per image size of the same workload through oracle/ (C + OpenMP), "
                                         "1 warm-up + median of 3"}
        sc_i = scenes.index(main_scene)
        ace_o, ace_t = solve_parity(main_scene, outs[sc_i][1], outs[sc_i][0], min(2, B))
        out["mean_corner_error_vs_ref_px"] = ace_o
        out["mean_corner_error_vs_truth_px"] = ace_t
    if stack_leg is not None:
        out["with_conv_stacks"] = stack_leg
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
