"""Resolution rules of compat/: FOUR reference modules are replaced, every other module of the checkout stays the checkout's.

    model.network, estimation, utils.kde, utils.local_correlation   ->  the files beside this one (re-exports of gfnet_amd)
    model.FPN, model.crossview_decoder_light, model.transformer.*, utils.utils, gfnet_configs, datasets, ...  ->  the user's checkout

Two mechanisms, because `sys.path` order alone cannot do it:
  * compat/model/__init__.py and compat/utils/__init__.py are regular packages that EXTEND their `__path__` with the checkout's
    `model/` and `utils/` directories (extend_package below), so `model.FPN` is found behind `model.network` -- compat/ in front of
    the checkout on PYTHONPATH is enough whenever the interpreter does not put the checkout first by itself;
  * `python test.py` / `python -m test` put the script's directory (the checkout) at sys.path[0], ahead of PYTHONPATH: the checkout's
    `estimation.py` and its regular `utils/` package would win.  install_finder() puts a meta-path finder in front of the path
    search that answers exactly the four names above from this directory; compat/run.py installs it and then runs the script.
Written for this repository; nothing of the reference is stored here.
"""
import importlib.abc
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
OVERRIDES = {
    "model.network": os.path.join(HERE, "model", "network.py"),
    "estimation": os.path.join(HERE, "estimation.py"),
    "utils.kde": os.path.join(HERE, "utils", "kde.py"),
    "utils.local_correlation": os.path.join(HERE, "utils", "local_correlation.py"),
}


def extend_package(name, path, exec_namespace=None):
    """Append to a compat package's __path__ every other `<entry>/<name>/` directory on sys.path (the checkout's), in sys.path order.
    A checkout package with a non-empty __init__.py has it executed in the compat package's namespace (the reference's are empty)."""
    own = os.path.realpath(os.path.join(HERE, name))
    for entry in list(sys.path):
        cand = os.path.join(entry or os.getcwd(), name)
        if not os.path.isdir(cand) or os.path.realpath(cand) == own or cand in path:
            continue
        path.append(cand)
        init = os.path.join(cand, "__init__.py")
        if exec_namespace is not None and os.path.isfile(init) and os.path.getsize(init) > 0:
            with open(init) as f:
                exec(compile(f.read(), init, "exec"), exec_namespace)
    return path


class _GfnetCompatFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        file = OVERRIDES.get(fullname)
        return importlib.util.spec_from_file_location(fullname, file) if file else None


def install_finder():
    """Idempotent.  Also makes `gfnet_amd` importable (the repository root goes on sys.path behind everything else)."""
    if not any(type(f).__name__ == "_GfnetCompatFinder" for f in sys.meta_path):  # (this file may be loaded under two names)
        sys.meta_path.insert(0, _GfnetCompatFinder())
    if REPO not in sys.path:
        sys.path.append(REPO)
