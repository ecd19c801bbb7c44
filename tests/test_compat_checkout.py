"""compat/ against a real checkout.  These tests run in the BUILD container only -- nothing of the reference
travels to the GPU box, so they skip when /root/reference is absent.  Each runs in a child process: `model` / `utils` / `estimation`
are top-level names that must not leak into the test process.

What must hold with compat/ in front of a checkout (reference test.py:9-11, 25-38; model/network.py:10-16):
  * model.network, estimation, utils.kde, utils.local_correlation are gfnet_amd's;
  * model.FPN, model.crossview_decoder_light, model.transformer, utils.utils are still the checkout's;
  * the reference's constructor call -- no backbone= argument -- gets the checkout's DINOv2 + decoder + FPN attached
    (GFNET_COMPAT_BACKBONE=reference), a full checkpoint loads into it, and the pyramids it hands the HIP path have the layout of
    GFNet.extract_features (network.py:156-201);
  * `python test.py` (checkout at sys.path[0]) is covered by compat/run.py.
"""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "model")), reason="no reference checkout in this environment")

# third-party modules the checkout imports at import time and this image lacks (torchvision, romatch): empty stand-ins on a
# directory of their own -- they are the user's installed packages in real use and take no part in what is asserted
STUBS = {
    "torchvision/__init__.py": "",
    "torchvision/transforms/__init__.py": "class _A:\n    def __init__(self, *a, **k): pass\nToTensor = Normalize = Resize = _A\n",
    "torchvision/transforms/functional.py": "import types\nInterpolationMode = types.SimpleNamespace(BICUBIC=3, BILINEAR=2)\n",
    "romatch/__init__.py": "",
    "romatch/utils/__init__.py": "",
    "romatch/utils/utils.py": "get_grid = get_autocast_params = None\n",
}

PRELUDE = """
import os, sys, json, warnings
import torch
"""


def _run(code, tmp_path, extra_env=None, cwd="/", argv=None, pythonpath=True):
    stubs = tmp_path / "stubs"
    for rel, text in STUBS.items():
        f = stubs / rel
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_text(text)
    env = dict(os.environ)
    env.pop("GFNET_COMPAT_BACKBONE", None)
    paths = [os.path.join(ROOT, "compat"), ROOT, REF, str(stubs)] if pythonpath else [str(stubs)]
    env["PYTHONPATH"] = os.pathsep.join(paths)
    env.update(extra_env or {})
    script = tmp_path / "child.py"
    script.write_text(PRELUDE + textwrap.dedent(code))
    r = subprocess.run(argv or [sys.executable, str(script)], env=env, capture_output=True, text=True, cwd=cwd, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    return r.stdout


def test_four_modules_are_overridden_and_the_rest_is_the_checkouts(tmp_path):
    out = _run("""
        import model.FPN, model.crossview_decoder_light, utils.utils, model.network, estimation, utils.kde, utils.local_correlation
        import gfnet_amd.model.network as n, gfnet_amd.estimation as e, gfnet_amd.utils.kde as k, gfnet_amd.utils.local_correlation as l
        assert model.network.GFNet is n.GFNet
        assert estimation.demo_estimation is e.demo_estimation and estimation.auc is e.auc
        assert utils.kde.kde is k.kde and utils.local_correlation.local_correlation is l.local_correlation
        ref = os.path.realpath("/root/reference")
        for m in (model.FPN, model.crossview_decoder_light, utils.utils):
            assert os.path.realpath(m.__file__).startswith(ref), m.__file__
        assert hasattr(model.FPN, "FPNEncoder") and hasattr(utils.utils, "get_tuple_transform_ops")
        from model.transformer import vit_large
        print("ok")
        """, tmp_path)
    assert out.strip().endswith("ok")


CONSTRUCT = """
    conf = json.load(open("/root/reference/gfnet_configs/basic.json"))
    conf["dino_cfg"]["decoder_cfg"]["attention_type"] = "FLASH2"

    class TinyViT(torch.nn.Module):  # stands in for DINOv2 ViT-L/14 (its 1.2 GB of weights are a download): same call, same token layout
        def __init__(self):
            super().__init__()
            self.proj = torch.nn.Conv2d(3, 1024, 14, 14)
        def forward_features(self, x):
            return {"x_norm_patchtokens": self.proj(x).flatten(2).transpose(1, 2)}

    import gfnet_amd.reference_backbone as rb
    rb._build_vit = lambda w: TinyViT().eval()
"""


def test_reference_constructor_call_gets_the_checkouts_backbone_and_a_full_checkpoint_loads(tmp_path):
    out = _run(CONSTRUCT + """
    from model.network import GFNet                      # test.py:9
    import model.FPN as fpn, model.crossview_decoder_light as cvd
    torch.manual_seed(0)
    model = GFNet(conf=conf, initial_res=(448, 448), upsample_res=(560, 560), symmetric=True, upsample_preds=True,
                  attenuate_cert=True)                   # test.py:25-30, no backbone argument
    bb = model.backbone
    assert type(bb.encoder) is fpn.FPNEncoder and type(bb.decoder) is fpn.FPNDecoder_concat
    assert type(bb.dino_decoder) is cvd.CrossVITDecoder_noself
    keys = set(model.state_dict())
    assert any(k.startswith("backbone.encoder.") for k in keys) and not any("dino.0" in k or "proj.weight" == k for k in keys)
    # a full reference checkpoint has the backbone's entries WITHOUT the "backbone." prefix (network.py:57-65)
    ckpt = {(k[len("backbone."):] if k.startswith("backbone.") else k): (v + 1 if v.is_floating_point() else v)
            for k, v in model.state_dict().items()}
    assert any(k.startswith("dino_decoder.") for k in ckpt) and any(k.startswith("merge_layer.") for k in ckpt)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                   # nothing set aside, so no warning
        res = model.load_state_dict(ckpt)                # test.py:38, strict
    assert not res.missing_keys and not res.unexpected_keys and model.ignored_backbone_keys == []
    for k, v in model.state_dict().items():
        assert torch.equal(v, ckpt[k[len("backbone."):] if k.startswith("backbone.") else k]), k
    print("ok")
    """, tmp_path, extra_env={"GFNET_COMPAT_BACKBONE": "reference"})
    assert out.strip().endswith("ok")


def test_checkpoint_before_backbone_is_not_lost(tmp_path):
    """construct -> load_state_dict -> assign model.backbone must end with the checkpoint's backbone weights."""
    out = _run(CONSTRUCT + """
    from model.network import GFNet
    import compat
    torch.manual_seed(0)
    donor = compat.reference_backbone(conf)
    ckpt = {k: torch.full_like(v, 0.25) if v.is_floating_point() else v for k, v in donor.state_dict().items()}
    model = GFNet(conf=conf)                              # GFNET_COMPAT_BACKBONE unset: no backbone yet
    assert model.backbone is None
    ckpt.update({k: v for k, v in model.state_dict().items()})
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        model.load_state_dict(ckpt)
    assert any("backbone entries" in str(x.message) for x in w)
    assert len(model.ignored_backbone_keys) == len(donor.state_dict())
    model.backbone = compat.reference_backbone(conf)      # assigned afterwards
    assert model.ignored_backbone_keys == [] and model.backbone_state == {}
    for k, v in model.backbone.state_dict().items():
        assert torch.equal(v, ckpt[k]), k
    print("ok")
    """, tmp_path)
    assert out.strip().endswith("ok")


def test_backbone_pyramids_have_the_layout_of_extract_features(tmp_path):
    """CPU forward of the assembled backbone on random weights (the decoder's FLASH2 attention is CUDA-only in the checkout,
    attention.py:230-231: the plain attention class stands in here -- an option of the same config key)."""
    out = _run(CONSTRUCT.replace('"FLASH2"', '"Linear"') + """
    import compat
    import model.transformer.layers.attention as att
    names = [n for n in ("Linear", "Attention", "MemEffAttention") if hasattr(att, n)]
    try:
        att.get_attention_type(conf["dino_cfg"]["decoder_cfg"]["attention_type"])
    except Exception:
        print("skip: no CPU attention class in the checkout"); sys.exit(0)
    torch.manual_seed(0)
    bb = compat.reference_backbone(conf, amp=False).eval()
    x = torch.randn(2, 3, 224, 224)
    try:
        with torch.no_grad():
            pa, pb = bb(x, False)
    except (NotImplementedError, AssertionError, RuntimeError) as e:
        print("skip: decoder attention not runnable on CPU:", str(e)[:80]); sys.exit(0)
    assert list(pa) == ["16", "8", "4", "2", "1"] and list(pb) == list(pa)
    want = {"16": (1, 64, 16, 16), "8": (1, 64, 28, 28), "4": (1, 32, 56, 56), "2": (1, 16, 112, 112), "1": (1, 8, 224, 224)}
    for k, shp in want.items():
        assert tuple(pa[k].shape) == shp and tuple(pb[k].shape) == shp, (k, pa[k].shape)
        assert pa[k].dtype == torch.float32 and pa[k].is_contiguous()
    with torch.no_grad():
        ua, ub = bb(x, True)
    assert list(ua) == ["8", "4", "2", "1"]
    print("ok")
    """, tmp_path)
    assert out.strip().endswith("ok"), out


def test_run_py_overrides_even_with_the_checkout_first_on_sys_path(tmp_path):
    """`python test.py` puts the checkout at sys.path[0], ahead of PYTHONPATH: its own estimation.py (cv2, kornia) and utils/ would
    win.  compat/run.py runs the unedited script with the four modules overridden.  A probe script INSIDE a copy-free stand-in of the
    checkout layout (a directory whose model/ utils/ estimation.py are symlinks to the checkout's) plays test.py's import lines."""
    co = tmp_path / "checkout"
    co.mkdir()
    for name in ("model", "utils", "estimation.py", "gfnet_configs"):
        os.symlink(os.path.join(REF, name), co / name)
    (co / "probe.py").write_text(textwrap.dedent("""
        import sys
        import gfnet_configs
        from model.network import GFNet
        from estimation import demo_estimation, auc
        import model.FPN, utils.utils, utils.kde
        import gfnet_amd.model.network as n, gfnet_amd.estimation as e, gfnet_amd.utils.kde as k
        assert GFNet is n.GFNet and demo_estimation is e.demo_estimation and utils.kde.kde is k.kde
        assert "checkout" in model.FPN.__file__ or "/root/reference" in model.FPN.__file__
        assert __name__ == "__main__" and sys.argv[1:] == ["--dataset", "mscoco"]
        print("ok")
        """))
    out = _run("", tmp_path, cwd=str(co), pythonpath=False,
               argv=[sys.executable, os.path.join(ROOT, "compat", "run.py"), "probe.py", "--dataset", "mscoco"])
    assert out.strip().endswith("ok")
    out = _run("", tmp_path, cwd=str(co), pythonpath=False,
               argv=[sys.executable, os.path.join(ROOT, "compat", "run.py"), "-m", "probe", "--dataset", "mscoco"])
    assert out.strip().endswith("ok")
