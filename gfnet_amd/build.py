"""Build libgfnet_hip.so (hand-written HIP for gfx950) in-tree: python -m gfnet_amd.build

hipcc cross-compiles without a GPU.  The library has a plain C ABI (include/gfnet_hip.h) and links
only against the HIP runtime; it is loaded with ctypes by gfnet_amd/_lib.py.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libgfnet_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         # no implicit FMA contraction: coordinate arithmetic must round exactly like the reference's
         # fp32 ops (FMAs in the kernels are explicit fmaf calls)
         "-ffp-contract=off",
         # the SLP vectoriser pairs independent fp32 FMA chains into v_pk_fma_f32 and pays for it in
         # v_mov shuffles and odd-sized LDS reads (measured: local-correlation D-stage 5x slower);
         # where packed math pays (kde.hip) it is written explicitly with vector types
         "-fno-slp-vectorize", "-fno-gpu-rdc"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps(src):
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "gfnet_hip.h"))
    return [src, os.path.abspath(__file__)] + hdrs


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


# per-file flags.  kde.hip: MFMA results straight into VGPRs -- every one of them feeds a v_exp_f32, which cannot read AGPRs
FILE_FLAGS = {"kde.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _compile(src, verbose, extra):
    obj = src[:-4] + ".o"
    extra = list(extra) + FILE_FLAGS.get(os.path.basename(src), [])
    if _stale(obj, _deps(src)):
        cmd = [HIPCC] + FLAGS + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj, True
    return obj, False


def build_ablate(verbose=False):
    """Timing-experiment build (tools/ only): same sources with -DGFN_ABLATE -> libgfnet_hip_ablate.so."""
    out = os.path.join(CSRC, "libgfnet_hip_ablate.so")
    cmd = [HIPCC] + FLAGS + ["-DGFN_ABLATE", "-shared", "-o", out] + sources()
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


def build_variant(name, flags, src_name="local_corr.hip", verbose=False):
    """A/B builds (tools/ab_lean.py): csrc/libgfnet_hip_<name>.so = the product objects with ONE source recompiled with extra flags.
    Not the product path."""
    build(verbose=verbose)
    src = os.path.join(CSRC, src_name)
    obj = os.path.join(CSRC, f"{src_name[:-4]}_{name}.o")
    out = os.path.join(CSRC, f"libgfnet_hip_{name}.so")
    cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(src_name, []) + list(flags) + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    others = [s[:-4] + ".o" for s in sources() if not s.endswith(src_name)]
    subprocess.run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", out, obj] + others, check=True)
    return out


def build(force=False, verbose=False, extra=()):
    srcs = sources()
    if force:
        for s in srcs:
            o = s[:-4] + ".o"
            if os.path.exists(o):
                os.remove(o)
    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, verbose, list(extra)), srcs))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    if "--ablate" in sys.argv:
        print(build_ablate(verbose=True))
        sys.exit(0)
    if "--variant" in sys.argv:  # python -m gfnet_amd.build --variant NAME [--src file.hip] <compiler flags, passed through verbatim>
        i = sys.argv.index("--variant")
        rest = sys.argv[i + 2:]
        src_name = "local_corr.hip"
        if "--src" in rest:
            j = rest.index("--src")
            src_name = rest[j + 1]
            rest = rest[:j] + rest[j + 2:]
        print(build_variant(sys.argv[i + 1], rest, src_name, verbose=True))  # (`-mllvm <opt>` keeps its value token: ADVICE r5)
        sys.exit(0)
    extra = [a for a in sys.argv[1:] if a.startswith("-") and a != "--force"]
    print(build(force="--force" in sys.argv, verbose=True, extra=extra))
