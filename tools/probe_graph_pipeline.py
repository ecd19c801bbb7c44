"""Pipelined hipGraph replay (Scene.capture_pipelined) against the eager step and the one-graph replay: bits and rate.
   python tools/probe_graph_pipeline.py WORKLOAD [steps]      (WORKLOAD: a key of gfnet_amd._synthetic.WORKLOADS)"""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gfnet_amd._synthetic import WORKLOADS, Scene  # noqa: E402

key = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
only = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else None  # sizes that get the two-stage form (the others: one graph)
wl = WORKLOADS[key]
dev = torch.device("cuda", 0)
dtype = torch.float16 if wl["dtype"] == "fp16" else torch.float32
scenes = [Scene(S, wl["pairs"], wl["num_itr"], dtype, "off", dev, 0) for S in wl["sizes"]]
pairs = wl["pairs"] * len(scenes)
with torch.inference_mode():
    ref = []
    for sc in scenes:
        for _ in range(2):
            H, good = sc.step(0)
        torch.cuda.synchronize()
        ref.append((H.clone(), good.clone()))
    print("eager ok", flush=True)
    piped = [sc for sc in scenes if only is None or sc.size in only]
    single = [sc for sc in scenes if sc not in piped]
    for sc in piped:
        sc.capture_pipelined(0)
    for sc in single:
        sc.capture(0)
    torch.cuda.synchronize()
    print("captured", flush=True)

    def one_step():
        for sc in scenes:
            if sc in piped:
                sc.replay_pipelined()
            else:
                sc.replay()

    for _ in range(4):
        one_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{key}: two-stage graphs for {[sc.size for sc in piped]} {pairs * steps / dt:.0f} pairs/s, {dt / steps * 1e3:.3f} ms per step", flush=True)
    for sc in scenes:
        sc.capture(0)
    torch.cuda.synchronize()
    for _ in range(2):
        for sc in scenes:
            sc.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for sc in scenes:
            sc.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{key}: one graph per scene {pairs * steps / dt:.0f} pairs/s, {dt / steps * 1e3:.3f} ms per step", flush=True)
    for sc in scenes:  # every scene's graph alone: its dependent chain of kernels
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            sc.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"   size {sc.size} alone: {dt / steps * 1e3:.3f} ms per step", flush=True)
    for sc in scenes:  # the host's share of a replay: time to enqueue one graph launch with an idle queue
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sc.replay()
            ts.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
        print(f"   size {sc.size}: host time of one graph launch {min(ts) * 1e3:.3f} ms (min of 5)", flush=True)
