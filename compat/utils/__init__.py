"""`utils` with compat/ in front on sys.path: `utils.kde` and `utils.local_correlation` are compat's (the HIP kernels), `utils.utils`
(reference model/network.py:12) and anything else come from the checkout's `utils/` directory, appended to this package's search
path (compat/_shim.py)."""
import importlib.util
import os

_spec = importlib.util.spec_from_file_location("_gfnet_compat_shim", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_shim.py"))
_shim = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_shim)
_shim.extend_package("utils", __path__, globals())
