"""Registers / spills / LDS of the kernels in a hipcc object or library: python tools/kernel_regs.py FILE [substring]
(carves the gfx950 code object out of the offload bundle and reads its metadata notes with llvm-readelf)"""
import re
import struct
import subprocess
import sys
import tempfile

data = open(sys.argv[1], "rb").read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
magic = b"__CLANG_OFFLOAD_BUNDLE__"
pos = data.find(magic)
while pos >= 0:
    n = struct.unpack_from("<Q", data, pos + 24)[0]
    off = pos + 32
    for _ in range(n):
        o, sz, tl = struct.unpack_from("<QQQ", data, off)
        triple = data[off + 24:off + 24 + tl].decode()
        off += 24 + tl
        if "gfx950" in triple and sz:
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(data[pos + o:pos + o + sz])
                f.flush()
                notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk)
                if not name or sub not in name.group(1):
                    continue
                g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [None, "?"])[1]
                dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
                print(f"{dem[:110]:110s} vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>4s} sgpr {g('sgpr_count'):>4s} "
                      f"sspill {g('sgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
    pos = data.find(magic, pos + 1)
