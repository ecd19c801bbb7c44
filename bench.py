#!/usr/bin/env python3
"""bench.py -- pairs/sec of the GFNet post-backbone hot path on MI355X (BASELINE.json metric).

One "step" = one batch of synthetic 448x448 image pairs (default 32 per GPU: BASELINE configs[1])
through the whole path on feature pyramids that are already resident in HBM:
  global correlation + soft-argmax (scale 16) -> per scale: refiner input (2 gathers, displacement
  embedding, local correlation into the concat buffer) + flow update + inter-scale resize, for the
  448 pass and the 560 refinement pass (test.py defaults: symmetric, upsample_preds,
  attenuate_cert) -> match post-processing -> balanced sampling (2 multinomials + KDE, N=M=20000)
  -> device RANSAC/DLT/LM homography solve -> (N>1) RCCL all-gather of the 3x3 matrices.
Excluded, stated in `config`: the DINOv2/FPN backbone and the refiners' conv stacks (PyTorch-ROCm
host code) -- the conv output (flow/certainty increment) is a zero tensor here.

    python bench.py [--gpus N] [--steps K] [--warmup W]        (N>1: launched by torchrun, one rank per GPU)

Prints ONE JSON line on rank 0 (see the driver contract); `roofline` is for the dominant kernel of
the named config, the scale-4 local-correlation launch (c32, 112^2, G64, r=4, 64 directions), timed
with HIP events on its launch stream inside the timed steps; `cpu_baseline` is the C/OpenMP oracle
(a port of the reference's algorithm, validated against it) on a bounded sample, rank 0, N=1 only.
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

S0, S1 = 448, 560
FEAT = {"16": 64, "8": 64, "4": 32, "2": 16, "1": 8}
DISP = {"16": 64, "8": 64, "4": 32, "2": 16, "1": 8}
CONF = {"encoder_cfg": {"feat_chs": [64, 32, 16, 8]},
        "matcher": {"num_grid": [32, 32, 64, 128, 256], "radius": [7, 6, 4, 2, 0],
                    "displacement_dim": [64, 64, 32, 16, 8], "num_itr": [1, 1, 1, 1, 1]}}  # gfnet_configs/basic.json
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
ROOFLINE_KEY = "local_corr_c32_h112_g64_r4"


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


def make_pyramids(H, size, scales, device, gen):
    """Synthetic feature pyramids: B-image features = smoothed noise (amplitude 2), A-image features =
    the B features seen through H + 0.1 noise, so correlation peaks and flows are meaningful."""
    n = H.shape[0]
    pa, pb = {}, {}
    for s in scales:
        side, c = (size // 14 if s == "16" else size // int(s)), FEAT[s]  # network.py:185-198: 32/56/112/224/448
        fb = F.avg_pool2d(torch.randn(n, c, side, side, device=device, generator=gen), 3, 1, 1) * 6.0
        fa = F.grid_sample(fb, warp_grid(H, side, size, device), mode="bilinear", padding_mode="zeros", align_corners=False)
        fa = fa + 0.1 * torch.randn(n, c, side, side, device=device, generator=gen)
        pa[s], pb[s] = fa.contiguous(), fb.contiguous()
    return pa, pb


class StandInRefiner(nn.Module):
    """The HIP part of ConvRefiner.forward (network.py:533-558) followed by a stand-in for the conv
    stack (network.py:560-563; PyTorch-ROCm/MIOpen, out of scope): like a trained refiner it returns
    the increment that moves the flow onto the true warp (smooth, also outside the overlap) and a
    constant certainty increment.  One tiny torch elementwise op; everything else is the real path."""

    def __init__(self, feat, disp, radius, scale, gt, conv_stack="off"):
        super().__init__()
        from gfnet_amd.model.network import ConvRefiner, _refiner_for

        K = (2 * radius + 1) ** 2 if radius > 0 else 0
        dim = 2 * feat + disp + K
        if conv_stack == "off":
            self.inner = ConvRefiner(dim, dim, 3, kernel_size=5, dw=True, hidden_blocks=0, displacement_emb="linear",
                                     displacement_emb_dim=disp, local_corr_num=radius, corr_in_other=radius > 0)
        else:
            # --conv-stack: the reference's refiner architecture for this scale (9 depthwise+1x1 blocks and the
            # output conv, random-init, eval) runs on the HIP conv-stack kernels.  Random weights
            # cannot refine anything, so its output enters with weight 0 and the stand-in increment below keeps the
            # synthetic scene consistent; all of its arithmetic is executed inside the timed region.
            self.inner = _refiner_for(feat, disp, radius)
            self.inner.conv_precision = conv_stack
        self.conv_stack = conv_stack
        self.scale, self.gt = scale, gt  # gt: {num_grid: (true flow (2B,2,G,G), image size)}
        self._cert, self._gtk = {}, {}

    def forward(self, num_grid, x, y, flow, scale_factor=1):
        d, lc = self.inner.assemble(num_grid, x, y, flow, scale_factor)
        gt, size = self.gt[num_grid]
        k = 4.0 * size / self.scale                       # undone by network.py:262-263's scale/(4*W0)
        if num_grid not in self._gtk:
            self._gtk[num_grid] = gt * k
        delta = torch.add(self._gtk[num_grid], flow, alpha=-k)  # (gt - flow) * k in one launch
        if num_grid not in self._cert:
            self._cert[num_grid] = torch.full((flow.shape[0], 1, num_grid, num_grid), 1.0, device=flow.device)
        cert = self._cert[num_grid]
        if self.conv_stack != "off":
            out = self.inner.conv_stack(d)
            delta = torch.addcmul(delta, out[:, :2], torch.zeros((), device=d.device))
            cert = torch.addcmul(cert, out[:, 2:3], torch.zeros((), device=d.device))
        return delta, cert, lc


def algorithmic_bytes_local_corr(B, c, hs, G, r):
    """f0 + f1 + flow + out, fp32: 3 489 792 B per pair-direction at scale 4."""
    return 4 * B * (c * G * G + c * hs * hs + 2 * G * G + (2 * r + 1) ** 2 * G * G)


def cpu_baseline(model, pyr, pyr_up, n_sample, seed_matches, sizes, gts, nb):
    """The same stages through the C/OpenMP oracle on `n_sample` pairs (host cores)."""
    import oracle

    scales = list(pyr[0].keys())
    t0 = time.time()
    for b in range(n_sample):
        def run_pass(p0, p1, size, grids, radii, scl, pre=None, sf=1.0):
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
                ref = model.conv_refiner[s].inner
                oracle.refiner_input(grids[i], f0[s], f1[s], flow, ref.disp_emb.weight.detach().cpu().numpy(),
                                     ref.disp_emb.bias.detach().cpu().numpy(), radii[i], scale_factor=sf,
                                     corr_in_other=radii[i] > 0)
                g_true = gts[grids[i]][[b, b + nb]]
                dl = (g_true - flow) * np.float32(4.0 * size / int(s))
                flow, cert, _ = oracle.flow_update(flow, cert, dl, np.ones_like(cert), np.full_like(flow, 1e-7), int(s), size, size)
                res[s] = (flow, cert)
                if s != "1":
                    flow = oracle.interpolate_bilinear(flow, grids[i + 1])
                    cert = oracle.interpolate_bilinear(cert, grids[i + 1])
            return res
        r1 = run_pass(pyr[0], pyr[1], S0, model.num_grid, model.radius, scales)
        gu, ru, _ = model.upsample_grids(S1)
        r2 = run_pass(pyr_up[0], pyr_up[1], S1, gu, ru, scales[1:], pre=r1["1"], sf=math.sqrt(S1 * S1 / (S0 * S0)))
        warp, cert = oracle.match_post(r2["1"][0], r2["1"][1], r1["16"][1], symmetric=True, attenuate_cert=True)
        torch.manual_seed(1234 + b)
        good, _ = oracle.sample(warp[0], cert[0], num=5000, device_is_gpu=True)
        pts = oracle.convert_matches(good, *sizes)
        oracle.homography_ransac(pts[None], thresh=3.0, iters=2000, seed=b)
    dt = time.time() - t0
    # parity of the solve on identical (GPU-sampled) matches: corner error between device H and oracle H
    pts = oracle.convert_matches(seed_matches["matches"], *sizes)
    Ho, _, _ = oracle.homography_ransac(pts, thresh=3.0, iters=2000, seed=0)
    err = [oracle.corner_error(Ho[i], seed_matches["H"][i], S0, S0, clamp=1e9) for i in range(len(Ho))]
    return n_sample / dt, float(np.mean(err)), oracle.max_threads()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs-per-gpu", type=int, default=32)
    ap.add_argument("--no-upsample", action="store_true", help="448 pass only (no 560 refinement pass)")
    ap.add_argument("--cpu-pairs", type=int, default=2, help="pairs for the CPU-oracle baseline leg (0 = skip)")
    ap.add_argument("--breakdown", action="store_true", help="print per-stage GPU times of one step to stderr")
    ap.add_argument("--conv-stack", choices=("off", "fp32", "fp16"), default="off",
                    help="also run the refiners' conv stacks (reference architecture, random-init) on the HIP kernels, with "
                         "fp32 or fp16 1x1-conv operands; default off = the north-star hot path only")
    args = ap.parse_args()

    from gfnet_amd import ops, parallel
    from gfnet_amd.estimation import estimate_homographies
    from gfnet_amd.model.network import GFNet, sample_batched

    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the hot path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B = args.pairs_per_gpu
    upsample = not args.no_upsample

    # ---- synthetic workload, resident in HBM before the timed region -------------------------------
    gen_cpu = torch.Generator().manual_seed(1000 + rank)
    gen = torch.Generator(device=dev).manual_seed(2000 + rank)
    H = random_homographies(B, S0, gen_cpu)
    scales = ["16", "8", "4", "2", "1"]
    pyr = make_pyramids(H, S0, scales, dev, gen)
    Hup = np.stack([np.diag([S1 / S0, S1 / S0, 1.0]) @ h @ np.diag([S0 / S1, S0 / S1, 1.0]) for h in H])
    pyr_up = make_pyramids(Hup, S1, scales[1:], dev, gen) if upsample else (None, None)
    # true normalised warps on every grid the two passes use (A->B for the first B rows, B->A after)
    Hinv, Hupinv = np.linalg.inv(H), (np.linalg.inv(Hup) if upsample else None)
    gt = {}
    for G in CONF["matcher"]["num_grid"]:
        gt[G] = (torch.cat((warp_grid(H, G, S0, dev), warp_grid(Hinv, G, S0, dev))).permute(0, 3, 1, 2).contiguous(), S0)
    if upsample:
        for G in (40, 80, 160, 320):
            gt[G] = (torch.cat((warp_grid(Hup, G, S1, dev), warp_grid(Hupinv, G, S1, dev))).permute(0, 3, 1, 2).contiguous(), S1)
    refiners = nn.ModuleDict({s: StandInRefiner(FEAT[s], DISP[s], CONF["matcher"]["radius"][i], int(s), gt, args.conv_stack)
                              for i, s in enumerate(scales)})
    model = GFNet(CONF, symmetric=True, upsample_preds=upsample, attenuate_cert=True, conv_refiner=refiners).to(dev).eval()
    sizes = (S0, S0, S0, S0)

    def step(seed):
        warp, cert = model.match_pyramids(pyr[0], pyr[1], pyr_up[0], pyr_up[1], batched=True)
        good, _ = sample_batched(model, warp, cert, 5000)
        Hl = estimate_homographies(good, sizes, iters=model.ransac_iters, seed=seed)
        return parallel.gather_homographies(Hl), good, Hl

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.inference_mode():
        for i in range(args.warmup):
            step(i)
        sync()
        if args.breakdown and rank == 0:
            def timed(fn):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
                return r, e0.elapsed_time(e1)
            (warp, cert), t_match = timed(lambda: model.match_pyramids(pyr[0], pyr[1], pyr_up[0], pyr_up[1], batched=True))
            (gm, _), t_sample = timed(lambda: sample_batched(model, warp, cert, 5000))
            _, t_solve = timed(lambda: estimate_homographies(gm, sizes, iters=model.ransac_iters, seed=0))
            print(f"[breakdown] match(448+560) {t_match:.2f} ms | sample {t_sample:.2f} ms | solve {t_solve:.2f} ms", file=sys.stderr)
        ops.kernel_events = {ROOFLINE_KEY: []}
        t0 = time.perf_counter()
        for i in range(args.steps):
            Hall, good, Hl = step(0)
        sync()
        dt = time.perf_counter() - t0
    events = ops.kernel_events[ROOFLINE_KEY]
    ops.kernel_events = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    pairs_per_s = world * B * args.steps / dt

    kern_us = float(np.mean([a.elapsed_time(b) for a, b in events])) * 1e3 if events else float("nan")
    nbytes = algorithmic_bytes_local_corr(2 * B, 32, 112, 64, 4)
    achieved = nbytes / (kern_us * 1e-6) / 1e9 if events else float("nan")
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "local_corr_pmc.json")  # written from the rocprofv3 --pmc passes
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "image pairs/sec at 448x448 (post-backbone hot path: correlation -> flow -> sampling -> homography)",
        "value": round(pairs_per_s, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "448x448 batch=32 synthetic pairs per GPU, local_correlation radius 7/6/4/2 (BASELINE configs[1])",
                   "pairs_per_gpu": B, "symmetric": True, "upsample_pass_560": upsample, "attenuate_cert": True,
                   "stages": "corr_softargmax, refiner_input+local_corr x(4+3 scales), flow_update, resize, match_post, "
                             "sample(2 draws without replacement + KDE 20000^2), RANSAC(2000)+DLT+LM, H all-gather",
                   "excluded": ("DINOv2/FPN backbone and refiner conv stacks (PyTorch-ROCm host code); stand-in increment = "
                                "exact residual to the true warp (1 torch elementwise op per refiner call)") if args.conv_stack == "off"
                   else "DINOv2/FPN backbone (PyTorch-ROCm host code)",
                   "refiner_conv_stack": "off" if args.conv_stack == "off" else
                   f"reference architecture (9 dw5x5+BN+ReLU+1x1 blocks + out conv per refiner call, C=417/361/177/73/24), random-init, "
                   f"HIP conv_stack kernels, 1x1 operands {args.conv_stack}; output weighted 0 next to the stand-in increment",
                   "parallelism": f"pairs sharded over {world} GPU(s), RCCL all-gather of H only"},
        "roofline": {"bound": "hbm", "kernel": "gfn_local_corr_fwd call = local_corr_tile_kernel<4,2> + local_corr_irregular_kernel<4,2> "
                                               "(c32, 112x112, G64, r4, 64 directions)",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": round(kern_us, 2)},
    }
    if rank == 0 and world == 1 and args.cpu_pairs > 0:
        to_np = lambda p: {s: t[: args.cpu_pairs].float().cpu().numpy() for s, t in p.items()}  # noqa: E731
        seed_matches = {"matches": good[: args.cpu_pairs].cpu().numpy(), "H": Hl[: args.cpu_pairs].cpu().numpy()}
        up = (to_np(pyr_up[0]), to_np(pyr_up[1])) if upsample else None
        if upsample:
            gts = {G: t[0].cpu().numpy() for G, t in gt.items()}
            v, ace, cores = cpu_baseline(model, (to_np(pyr[0]), to_np(pyr[1])), up, args.cpu_pairs, seed_matches, sizes, gts, B)
            out["cpu_baseline"] = {"value": round(v, 4), "unit": "pairs/s", "cores": cores, "kind": "port",
                                   "sample": f"{args.cpu_pairs} pairs of the same workload through oracle/ (C + OpenMP)"}
            out["mean_corner_error_vs_ref_px"] = ace
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
