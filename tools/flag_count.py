import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch, synth
from gfnet_amd import _lib
from gfnet_amd.utils.local_correlation import local_correlation
B = 64
for (c, hs, G, r) in [(32, 112, 64, 4), (16, 224, 128, 2), (32, 140, 80, 4), (16, 280, 160, 2)]:
    f0 = torch.randn(B, c, G, G, device="cuda"); f1 = torch.randn(B, c, hs, hs, device="cuda")
    flow = torch.from_numpy(np.tile(synth.homography_flow(2, G, 5), (B // 2, 1, 1, 1))).cuda()
    out = torch.empty(B, (2 * r + 1) ** 2, G, G, device="cuda")
    for v in (0, 4):
        local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow, out=out, _variant=v)
        torch.cuda.synchronize()
        dev = torch.device("cuda", 0)
        hdr = _lib.scratch(dev, int(_lib.lib().gfn_local_corr_scratch_bytes(B, G)))[:8].cpu().numpy()
        tiles = B * ((G + 3) // 4) * ((G + 15) // 16)
        print(f"c{c} hs{hs} G{G} r{r} variant {v}: tiles {tiles}, second-launch tiles {hdr[3]}, cells redone per tap {hdr[5]} ({hdr[5] / (B * G * G):.2e} of cells), halves tiles ~{hdr[7]}", flush=True)
