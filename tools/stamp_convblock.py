"""Phase time stamps of a few workgroups of the fused conv block on half maps (GFN_ABLATE build, device printf).
Stamps (s_memtime ticks, 100 MHz): 1 first loads issued, 2 halo zeroed; per K tile t: 3+6t committed, 4+6t barrier, 5+6t next loads
issued, 6+6t depthwise done, 7+6t barrier, 8+6t matrix step done; 34.. item stored."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GFNET_HIP_LIB"] = os.path.join(ROOT, "gfnet_amd", "csrc", "libgfnet_hip_ablate.so")
sys.path.insert(0, ROOT)
import torch
from gfnet_amd import _lib, ops
from gfnet_amd._lib import ptr, stream_ptr
C, G = (24, 256) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
B = 64
packed = ops.conv_block_pack(torch.randn(C, 25, device="cuda") * 0.2, torch.randn(C, device="cuda"), torch.rand(C, device="cuda") + 0.5,
                             torch.randn(C, device="cuda"), torch.randn(C, C, device="cuda") * C ** -0.5, torch.randn(C, device="cuda"))
xh = torch.randn(B, (C + 1) // 2, G, G, 2, device="cuda").half()
yh = torch.empty_like(xh)
extra = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # more phases switched off (tools/ablate_convblock.py masks)
for m in (extra, extra, 64 | extra):
    _lib.lib().gfn_conv_block_half_fwd(ptr(xh), (m << 8) | 1, ptr(packed), ptr(yh), 1, B, C, C, G, stream_ptr(xh.device))
    torch.cuda.synchronize()

# the build prints one line per stamp: "CS wg wave index ticks"; run as  python tools/stamp_convblock.py C G | python tools/stamp_convblock.py --table
