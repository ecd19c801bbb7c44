"""`model` with compat/ in front on sys.path: `model.network` is compat's (gfnet_amd), every other submodule -- `model.FPN`,
`model.crossview_decoder_light`, `model.transformer` (reference model/network.py:14-15, :47) -- comes from the checkout's `model/`
directory, which this package appends to its search path (compat/_shim.py)."""
import importlib.util
import os

_spec = importlib.util.spec_from_file_location("_gfnet_compat_shim", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_shim.py"))
_shim = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_shim)
_shim.extend_package("model", __path__, globals())
