"""A/B of library builds on the lean local-correlation shapes (GPU): python tools/ab_lean.py LIB [LIB ...]
Every library runs in a child process of its own (GFNET_HIP_LIB), interleaved over REPS rounds so that clock drift hits all of them
alike; per shape the table gives the median microseconds of the C-ABI call (plan launch included, as tools/probe_local_corr.py) and a
checksum of the output (equal checksums = bit-identical results).  Not part of the product."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(32, 112, 64, 4), (32, 140, 80, 4), (16, 224, 128, 2), (16, 280, 160, 2), (64, 56, 32, 6), (64, 32, 32, 7)]


def child():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import numpy as np
    import torch
    import synth
    from gfnet_amd.utils.local_correlation import local_correlation
    B = 64
    kind = os.environ.get("AB_FLOW", "homography")
    shapes = [SHAPES[int(i)] for i in os.environ.get("AB_SHAPES", "0,1,2,3").split(",")]
    res = {}
    for (c, hs, G, r) in shapes:
        g = torch.Generator().manual_seed(1)
        f0 = torch.randn(B, c, G, G, generator=g).cuda()
        f1 = torch.randn(B, c, hs, hs, generator=g).cuda()
        if kind == "homography":
            flow = torch.from_numpy(np.tile(synth.homography_flow(2, G, 5), (B // 2, 1, 1, 1))).cuda()
        elif kind == "bench":  # the bench's flows: true warps of 15 % corner-perturbation homographies (both directions) + 0.5-px noise
            from gfnet_amd import _synthetic as synthetic
            S = 4 * hs if r == 4 else (2 * hs if r == 2 else 8 * hs)
            Hm = synthetic.random_homographies(B // 2, S, g)
            flow = torch.cat((synthetic.warp_grid(Hm, G, S, "cpu"), synthetic.warp_grid(np.linalg.inv(Hm), G, S, "cpu"))).permute(0, 3, 1, 2)
            flow = (flow + torch.randn(B, 2, G, G, generator=g) * (0.5 / S)).contiguous().cuda()
        elif kind == "sprinkle":  # the bench's situation: smooth flows, a handful of tiles per launch left to the second launch
            flow = torch.from_numpy(np.tile(synth.homography_flow(2, G, 5), (B // 2, 1, 1, 1)))
            for k in range(3):
                b, i, j = (int(v) for v in torch.randint(0, min(B, G), (3,), generator=g))
                flow[b % B, :, i % G, j % G] = torch.rand(2, generator=g) * 1.8 - 0.9
            flow = flow.cuda()
        else:
            flow = (torch.rand(B, 2, G, G, generator=g) * 1.8 - 0.9).cuda()
        out = torch.empty(B, (2 * r + 1) ** 2, G, G, device="cuda")
        for _ in range(3):
            local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow, out=out)
        torch.cuda.synchronize()
        n = 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow, out=out)
        e1.record()
        torch.cuda.synchronize()
        h = hashlib.md5(out.cpu().numpy().tobytes()).hexdigest()[:8]
        res[f"c{c} {hs}^2 G{G} r{r}"] = (e0.elapsed_time(e1) * 1e3 / n, h)
    print("AB_RESULT " + json.dumps(res), flush=True)


def main():
    libs = sys.argv[1:]
    reps = int(os.environ.get("AB_REPS", "3"))
    table = {}
    for rep in range(reps):
        for lib in libs:
            env = dict(os.environ, GFNET_HIP_LIB=os.path.abspath(lib), AB_CHILD="1")
            p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("AB_RESULT ")]
            if not line:
                print(f"{lib}: FAILED\n{p.stdout[-2000:]}\n{p.stderr[-3000:]}", flush=True)
                continue
            for shape, (us, h) in json.loads(line[0][10:]).items():
                table.setdefault(shape, {}).setdefault(lib, []).append((us, h))
    for shape, per in table.items():
        print(shape)
        for lib in libs:
            v = per.get(lib, [])
            if v:
                us = sorted(u for u, _ in v)
                print(f"    {os.path.basename(lib):40s} median {us[len(us) // 2]:8.1f} us  (min {us[0]:.1f}, max {us[-1]:.1f})  md5 {v[0][1]}"
                      + ("" if len({h for _, h in v}) == 1 else "  CHECKSUM VARIES"), flush=True)


if __name__ == "__main__":
    if os.environ.get("AB_CHILD"):
        child()
    else:
        main()
