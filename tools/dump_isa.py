"""Disassembly of the gfx950 kernels in a hipcc object or library: python tools/dump_isa.py FILE OUT.s
(carves the code object out of the offload bundle like tools/kernel_regs.py, then llvm-objdump -d; names demangled)"""
import struct
import subprocess
import sys

data = open(sys.argv[1], "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
pos = data.find(magic)
n_out = 0
while pos >= 0:
    n = struct.unpack_from("<Q", data, pos + 24)[0]
    off = pos + 32
    for _ in range(n):
        o, sz, tl = struct.unpack_from("<QQQ", data, off)
        triple = data[off + 24:off + 24 + tl].decode()
        off += 24 + tl
        if "gfx950" in triple and sz:
            co = sys.argv[2] + (".%d" % n_out if n_out else "") + ".co"
            open(co, "wb").write(data[pos + o:pos + o + sz])
            dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
            dis = subprocess.run(["c++filt"], input=dis, capture_output=True, text=True).stdout
            open(sys.argv[2] + (".%d" % n_out if n_out else ""), "w").write(dis)
            n_out += 1
    pos = data.find(magic, pos + 1)
print(n_out, "code objects")
