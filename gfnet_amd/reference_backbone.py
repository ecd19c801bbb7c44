"""The reference's feature extractor, built from the USER'S OWN GFNet checkout, as a `backbone=` for gfnet_amd's GFNet.

BASELINE north_star: "host code stays PyTorch-ROCm for the FPN/transformer backbone".  Nothing of that backbone lives in this
package; this module only *assembles* it from the checkout's classes when one is importable (`model.FPN`,
`model.crossview_decoder_light`, `model.transformer` -- with compat/ on sys.path those still resolve to the checkout, see
compat/model/__init__.py) and states the data flow of `GFNet.extract_features` (reference model/network.py:156-201) around them:

    DINOv2 ViT-L/14 patch tokens (frozen, amp dtype)  -> cross-view decoder -> stride-16 features (B, 64, H/14, W/14)
    FPN encoder on the images -> conv31 += merge_layer(cat(conv31, resized ViT features)) -> FPN decoder -> strides 8, 4, 2, 1
    pyramids = {"16": vit, "8": feat1, "4": feat2, "2": feat3, "1": feat4}, first half of the batch = image A, second = image B;
    the refinement pass (upsample=True) drops "16".

The submodule names are the reference's (`dino_decoder`, `encoder`, `decoder`, `merge_layer`: network.py:57-65), so the entries of a
full reference checkpoint route into them (GFNet.load_state_dict); the ViT hangs in a plain list like the reference's `self.dino`
(network.py:56: kept out of the state dict, moved to the device lazily).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

DINOV2_URL = "https://dl.fbaipublicfiles.com/dinov2/dinov2_vitl14/dinov2_vitl14_pretrain.pth"  # network.py:46


def checkout_available():
    """True when the reference's backbone modules can be found on sys.path (nothing is imported)."""
    import importlib.util

    try:
        return all(importlib.util.find_spec(m) is not None for m in ("model.FPN", "model.crossview_decoder_light"))
    except (ImportError, ValueError):
        return False


def _build_vit(dino_weights):
    from model.transformer import vit_large  # the checkout's DINOv2 (needs its own third-party imports, e.g. romatch)

    vit = vit_large(img_size=518, patch_size=14, init_values=1.0, ffn_layer="mlp", block_chunks=0).eval()  # network.py:48-54
    if dino_weights is None:
        dino_weights = os.environ.get("GFNET_DINOV2_WEIGHTS")
    if dino_weights is None:
        state = torch.hub.load_state_dict_from_url(DINOV2_URL, map_location="cpu")  # what the reference's constructor does
    elif isinstance(dino_weights, (str, os.PathLike)):
        state = torch.load(dino_weights, map_location="cpu")
    else:
        state = dino_weights
    vit.load_state_dict(state)
    return vit


class ReferenceBackbone(nn.Module):
    """`backbone(images, upsample) -> (pyramid_A, pyramid_B)` from the checkout's DINOv2 + CrossVITDecoder_noself + FPN.

    vit: an already built ViT (anything with `forward_features(x)["x_norm_patchtokens"]`); None builds the checkout's `vit_large`
    and loads `dino_weights` (a path, a state dict, $GFNET_DINOV2_WEIGHTS, or the reference's download URL in that order)."""

    def __init__(self, conf, amp=True, amp_dtype=torch.float16, vit=None, dino_weights=None):
        super().__init__()
        from model.crossview_decoder_light import CrossVITDecoder_noself
        from model.FPN import FPNDecoder_concat, FPNEncoder, Swish

        self.amp, self.amp_dtype = amp, amp_dtype
        if vit is None:
            vit = _build_vit(dino_weights)
        for p in vit.parameters():
            p.requires_grad = False
        self.dino = [vit]
        chs = list(conf["encoder_cfg"]["feat_chs"])  # coarse to fine
        self.dino_decoder = CrossVITDecoder_noself(conf=conf, upsample=False)
        self.encoder = FPNEncoder(feat_chs=chs[::-1])
        self.decoder = FPNDecoder_concat(feat_chs=chs[::-1])
        self.merge_layer = nn.Sequential(nn.Conv2d(2 * chs[0], chs[0], kernel_size=3, padding=1), nn.BatchNorm2d(chs[0]), Swish())

    def _vit_tokens(self, x):
        vit = self.dino[0]
        p = next(vit.parameters(), None)
        if p is not None and (p.device != x.device or (x.is_cuda and p.dtype != self.amp_dtype)):
            vit = self.dino[0] = vit.to(x.device).to(self.amp_dtype if x.is_cuda else p.dtype)
            p = next(vit.parameters())
        with torch.no_grad():
            return vit.forward_features(x.to(p.dtype) if p is not None else x)["x_norm_patchtokens"]

    def forward(self, x, upsample=False):
        n2, ch, H, W = x.shape
        hv, wv = H // 14 * 14, W // 14 * 14
        tokens = self._vit_tokens(x if (H, W) == (hv, wv) else F.interpolate(x, (hv, wv), mode="bilinear", align_corners=False))
        ta, tb = tokens.chunk(2)
        with torch.autocast(device_type="cuda", enabled=bool(self.amp) and x.is_cuda, dtype=self.amp_dtype):
            va, vb = self.dino_decoder(ta, tb, vit_shape=(n2 // 2, ch, hv // 14, wv // 14))
        vit_feat = torch.cat((va.float(), vb.float()))
        c0, c1, c2, c3 = self.encoder(x)
        side = vit_feat if tuple(vit_feat.shape[2:]) == (H // 8, W // 8) else \
            F.interpolate(vit_feat, size=(H // 8, W // 8), mode="bilinear", align_corners=False)
        c3 = c3 + self.merge_layer(torch.cat((c3, side), dim=1))
        levels = [vit_feat] + list(self.decoder(c0, c1, c2, c3))
        pyr_a, pyr_b = {}, {}
        for name, t in zip(("16", "8", "4", "2", "1"), levels):
            if upsample and name == "16":
                continue
            a, b = t.float().chunk(2)
            pyr_a[name], pyr_b[name] = a.contiguous(), b.contiguous()
        return pyr_a, pyr_b


def reference_backbone(conf, amp=True, amp_dtype=torch.float16, vit=None, dino_weights=None):
    """Build the backbone from the checkout on sys.path; raises ImportError with the reason when there is none."""
    if not checkout_available():
        raise ImportError("no KN-Zhang/GFNet checkout on sys.path: `model.FPN` / `model.crossview_decoder_light` cannot be found "
                          "(put the checkout behind compat/ on PYTHONPATH, or run the script through compat/run.py)")
    return ReferenceBackbone(conf, amp=amp, amp_dtype=amp_dtype, vit=vit, dino_weights=dino_weights)
