"""Soak of the three-stage streaming at production size (GPU box): python tools/soak_stages.py [steps] [workload]
   N steps of the workload on one stream (reference), then the same N steps -- same seeds, same generator state -- through
   bench.SceneRunner's three stages with all of them in flight, every step's H and sampled matches compared bit for bit."""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from gfnet_amd._synthetic import WORKLOADS, Scene  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
key = sys.argv[2] if len(sys.argv) > 2 else "448b32"
wl = WORKLOADS[key]
dev = torch.device("cuda", 0)
sc = Scene(wl["sizes"][0], wl["pairs"], wl["num_itr"], torch.float32, "off", dev, 0)
with torch.inference_mode():
    torch.manual_seed(123)
    ref = []
    for i in range(steps):
        H, good = sc.step(i)
        ref.append((H.clone(), good.clone()))
    torch.cuda.synchronize()
    runner = bench.SceneRunner([sc], pipeline=True, stages=3)
    torch.manual_seed(123)
    outs = [runner.step(i)[0] for i in range(steps)]
    torch.cuda.synchronize()
    bad = [i for i, ((H, g), (Hr, gr)) in enumerate(zip(outs, ref)) if not (torch.equal(H, Hr) and torch.equal(g, gr))]
    print(f"{key}: {steps} steps in three stages, {len(bad)} differ from the one-stream steps", bad[:10])
    sys.exit(1 if bad else 0)
