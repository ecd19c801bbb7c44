"""gfnet_amd -- MI355X-native hot path of GFNet (grid-based dense correlation -> flow ->
balanced sampling -> homography solve) behind the reference's own Python surface.

Layout mirrors the reference for the functions on the path:
  gfnet_amd.utils.local_correlation.local_correlation   <- utils/local_correlation.py
  gfnet_amd.utils.kde.kde                                <- utils/kde.py
  gfnet_amd.model.network                                <- model/network.py (hot-path part)
  gfnet_amd.estimation                                   <- estimation.py
All arithmetic runs in csrc/*.hip (gfx950) through the C ABI in include/gfnet_hip.h.
"""
__version__ = "0.1.0"
