"""`from utils.local_correlation import local_correlation` (reference model/network.py:11) -> the HIP kernel behind the same signature."""
from gfnet_amd.utils.local_correlation import local_correlation  # noqa: F401
