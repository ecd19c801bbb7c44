"""Which streams of a process share a hardware queue (GPU box): python tools/probe_streams.py
   Nine fresh torch streams, pairwise overlap test of gfnet_amd.parallel (two one-workgroup spin kernels: 1 = they run side by side,
   0 = one after the other: the same hardware queue), then the pool parallel.concurrent_streams builds from such tests."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from gfnet_amd import parallel
cands = [torch.cuda.Stream() for _ in range(9)]
for i in range(9):
    print(i, "".join("1" if i != j and parallel._overlap(cands[i], cands[j]) else ("-" if i == j else "0") for j in range(9)))
pool = parallel.concurrent_streams(4)
print("pool of 4:", [[int(parallel._overlap(a, b)) for b in pool if b is not a] for a in pool])
pool6 = parallel.concurrent_streams(6)
print("pool of 6 (more than the queues):", len(pool6))
