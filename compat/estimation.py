"""`import estimation` of the reference (test.py:10-11, benchmark/multimodal_homog_benchmark_multiscale.py:5-6) -> gfnet_amd.estimation."""
from gfnet_amd.estimation import auc, convert_coordinates, corner_error, demo_estimation, estimate_homographies  # noqa: F401
