"""hipGraph layouts of the three-scene pyramid workload on the four hardware queues (round 4):
   python tools/probe_graph_layouts.py [steps]
   A: one graph per scene, the largest scene in two stages (4 streams)        -- bench.py's layout
   B: the largest scene in three stages, the two small scenes one after the other on the fourth stream
   C: the largest scene in three stages, the middle scene on the fourth stream, the small scene behind the largest's first stage
   F: as C with the small scene behind the largest's sampling + solve stage
   G: the two larger scenes in two stages each, the small scene behind the middle scene's second stage
   (measured: A 8.5 k pairs/s, B 6.6 k, C 7.5 k, F 7.7-7.8 k, G 7.9 k; the 672 scene alone 1.99 ms in two stages, 1.68 ms in three)"""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gfnet_amd import parallel  # noqa: E402
from gfnet_amd._synthetic import WORKLOADS, Scene  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
wl = WORKLOADS["pyr-fp16"]
dev = torch.device("cuda", 0)
NQ = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
pool = parallel.concurrent_streams(6 if NQ >= 8 else 4)


def scenes():
    return [Scene(S, wl["pairs"], wl["num_itr"], torch.float16, "off", dev, 0) for S in wl["sizes"]]


def timed(name, step):
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {3 * wl['pairs'] * steps / dt:.0f} pairs/s, {dt / steps * 1e3:.3f} ms per step", flush=True)


with torch.inference_mode():
    s224, s448, s672 = scenes()
    s224.capture(0, stream=pool[0])
    s448.capture(0, stream=pool[1])
    s672.capture_pipelined(0, streams=(pool[2], pool[3]))
    timed("A  224 | 448 | 672 in two stages", lambda: (s224.replay(), s448.replay(), s672.replay_pipelined()))
    del s224, s448, s672
    s224, s448, s672 = scenes()
    s224.capture(0, stream=pool[3])
    s448.capture(0, stream=pool[3])
    s672.capture_pipelined(0, streams=(pool[0], pool[1], pool[2]), stages=3)
    timed("B  672 in three stages | 224 then 448 on one stream", lambda: (s672.replay_pipelined(), s224.replay(), s448.replay()))
    del s224, s448, s672
    s224, s448, s672 = scenes()
    s224.capture(0, stream=pool[0])
    s448.capture(0, stream=pool[3])
    s672.capture_pipelined(0, streams=(pool[0], pool[1], pool[2]), stages=3)
    timed("C  672 in three stages (224 behind its first stage) | 448", lambda: (s672.replay_pipelined(), s224.replay(), s448.replay()))
    del s224, s448, s672
    s224, s448, s672 = scenes()
    s224.capture(0, stream=pool[2])
    s448.capture(0, stream=pool[3])
    s672.capture_pipelined(0, streams=(pool[0], pool[1], pool[2]), stages=3)
    timed("F  672 in three stages (224 behind its sampling + solve) | 448", lambda: (s672.replay_pipelined(), s224.replay(), s448.replay()))
    timed("F' the same, 224 enqueued first", lambda: (s224.replay(), s672.replay_pipelined(), s448.replay()))
    del s224, s448, s672
    s224, s448, s672 = scenes()
    s224.capture(0, stream=pool[3])
    s448.capture_pipelined(0, streams=(pool[2], pool[3]))
    s672.capture_pipelined(0, streams=(pool[0], pool[1]))
    timed("G  672 in two stages | 448 in two stages (224 behind its sampling + solve)", lambda: (s672.replay_pipelined(), s448.replay_pipelined(), s224.replay()))
    del s224, s448
    timed("   672 alone in two stages", lambda: s672.replay_pipelined())
    if NQ >= 8:
        del s672
        s224, s448, s672 = scenes()
        s224.capture(0, stream=pool[5])
        s448.capture_pipelined(0, streams=(pool[3], pool[4]))
        s672.capture_pipelined(0, streams=(pool[0], pool[1], pool[2]), stages=3)
        timed("D  672 in three stages | 448 in two | 224 (six streams, GPU_MAX_HW_QUEUES=8)", lambda: (s672.replay_pipelined(), s448.replay_pipelined(), s224.replay()))
        del s224, s448, s672
        s224, s448, s672 = scenes()
        s224.capture(0, stream=pool[4])
        s448.capture(0, stream=pool[3])
        s672.capture_pipelined(0, streams=(pool[0], pool[1], pool[2]), stages=3)
        timed("E  672 in three stages | 448 | 224 (five streams)", lambda: (s672.replay_pipelined(), s448.replay(), s224.replay()))
