"""`from utils.kde import kde` (reference model/network.py:10) -> the HIP kernel behind the same signature."""
from gfnet_amd.utils.kde import kde  # noqa: F401
