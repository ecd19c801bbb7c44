"""Compile the C oracle (test infrastructure) with gcc: python -m oracle.build

Outputs go to oracle/_build/ (git-ignored; they travel to the GPU box with the gpurun snapshot).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")
SOURCES = ["gfnet_oracle.c", "homography_oracle.c"]
VARIANTS = {"f32": "float", "f64": "double"}


def lib_path(variant):
    return os.path.join(BUILD, f"liboracle_{variant}.so")


def _stale(out, srcs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=False):
    os.makedirs(BUILD, exist_ok=True)
    srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    for variant, real in VARIANTS.items():
        out = lib_path(variant)
        if not force and not _stale(out, srcs + [os.path.abspath(__file__)]):
            continue
        # -ffp-contract=off: no silent FMA fusion, so the fp32 build keeps the reference's
        # one-rounding-per-op arithmetic; -fno-fast-math for the same reason.
        cmd = ["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-fno-fast-math",
               "-fvisibility=hidden", f"-DREAL={real}", "-o", out] + srcs + ["-lm"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return [lib_path(v) for v in VARIANTS]


if __name__ == "__main__":
    print("\n".join(build(force="--force" in sys.argv, verbose=True)))
