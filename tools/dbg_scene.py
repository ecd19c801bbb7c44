import os, sys, faulthandler
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
from gfnet_amd import _lib, ops
orig = _lib.check
def check(code, what):
    orig(code, what)
    print("ok launch", what, flush=True)
    if os.environ.get("DBG_SYNC") == "1":
        torch.cuda.synchronize()
        print("   synced", what, flush=True)
_lib.check = check
ops.check = check
import gfnet_amd.utils.local_correlation as lc
from gfnet_amd._synthetic import Scene
S, pairs = int(sys.argv[1]), int(sys.argv[2])
dt = torch.float16 if sys.argv[3] == "fp16" else torch.float32
dev = torch.device("cuda", 0)
sc = Scene(S, pairs, [1] * 5, dt, "off", dev, 0)
with torch.inference_mode():
    for i in range(int(os.environ.get("DBG_STEPS", "1"))):
        print("---- step", i, flush=True)
        H, g = sc.step(5)
    torch.cuda.synchronize()
    print("step ok", H.shape)
