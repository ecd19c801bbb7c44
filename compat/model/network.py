"""`from model.network import GFNet` (reference test.py:9) -> gfnet_amd.model.network (same constructor arguments, match / sample /
corr_volume / pos_embed surface, load_state_dict of a full reference checkpoint)."""
from gfnet_amd.model.network import ConvRefiner, GFNet, sample_batched  # noqa: F401
