"""`import compat` (repository root on sys.path): the shim's helpers.  The directory itself is what goes on sys.path for the
reference's module names -- see README.md beside this file."""
from gfnet_amd.reference_backbone import ReferenceBackbone, checkout_available, reference_backbone  # noqa: F401

from ._shim import OVERRIDES, install_finder  # noqa: F401
