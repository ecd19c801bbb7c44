"""Run one of the reference's scripts, unedited, on the MI355X path:

    cd /path/to/GFNet-checkout
    GFNET_COMPAT_BACKBONE=reference python /path/to/this/repo/compat/run.py test.py --dataset mscoco --conf_path gfnet_configs/basic.json --ckpt_path ...
    GFNET_COMPAT_BACKBONE=reference python /path/to/this/repo/compat/run.py -m benchmark.multimodal_homog_benchmark_multiscale ...

`python test.py` itself puts the checkout at sys.path[0], where its own `estimation.py` and `utils/` would win over PYTHONPATH; this
launcher installs compat's finder (compat/_shim.py: model.network, estimation, utils.kde, utils.local_correlation -> gfnet_amd,
everything else -> the checkout) and then runs the script or module as `__main__` with the path layout the interpreter would have
given it.  No GPU call happens before the script's own code runs.
"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _shim  # noqa: E402

sys.path.pop(0)


def main(argv):
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 2
    _shim.install_finder()
    if argv[0] == "-m":
        if len(argv) < 2:
            print("compat/run.py -m <module> [args]", file=sys.stderr)
            return 2
        sys.path.insert(0, os.getcwd())
        sys.argv = [argv[1]] + argv[2:]
        runpy.run_module(argv[1], run_name="__main__", alter_sys=True)
    else:
        script = os.path.abspath(argv[0])
        sys.path.insert(0, os.path.dirname(script))
        sys.argv = [script] + argv[1:]
        runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
