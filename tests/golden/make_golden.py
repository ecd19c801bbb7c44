#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Run in the build container only (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

The reference is imported unmodified from /root/reference with import-time stubs for the
third-party modules that are absent here (torchvision, romatch, cv2, kornia, kornia_moons);
none of the stubbed modules takes part in the arithmetic that is recorded.  Every fixture is
plain data (inputs + the reference's outputs) in an .npz; torch/numpy versions are recorded in
meta.json.  Fixture ids follow SURVEY.md section 8(c): G1..G8.
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402

REF = os.environ.get("GFNET_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            raise RuntimeError("stubbed third-party call")

    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms", ToTensor=_Any, Normalize=_Any, Resize=_Any)
    tvf = _stub("torchvision.transforms.functional", InterpolationMode=types.SimpleNamespace(BICUBIC=3))
    tv.transforms.functional = tvf
    _stub("romatch")
    _stub("romatch.utils")
    _stub("romatch.utils.utils", get_grid=None, get_autocast_params=None)
    cv2 = _stub("cv2", RANSAC=8)
    k = _stub("kornia")
    k.feature = _stub("kornia.feature")
    k.geometry = types.SimpleNamespace()
    _stub("kornia_moons")
    _stub("kornia_moons.viz", draw_LAF_matches=None)
    return cv2


def t2n(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path)/1024:.1f} KiB")


# ------------------------------------------------------------------------------------------
def g1_local_correlation(local_correlation):
    """utils/local_correlation.py:4-72"""
    g = torch.Generator().manual_seed(101)
    # (a) small non-square map, flow partly out of range (zeros padding), k ordering
    B, c, h, w, G, r = 2, 8, 20, 28, 6, 2
    f0 = torch.randn(B, c, G, G, generator=g)
    f1 = torch.randn(B, c, h, w, generator=g)
    flow = torch.rand(B, 2, G, G, generator=g) * 2.6 - 1.3
    out = local_correlation((B, c, h, w), f0, f1, local_radius=r, num_grid=G, flow=flow)
    save("g1a_local_corr_small", f0=t2n(f0), f1=t2n(f1), flow=t2n(flow), out=t2n(out),
         r=np.int64(r), G=np.int64(G))

    # (b) scale-4 shape (c32, 112^2, G64, r4), B=2: smooth homography-like flow + jitter, batch 1
    #     partly outside the image.  Inputs come from tests/golden/synth.py (regenerated, not stored);
    #     the fixture keeps 512 probe entries + full-tensor checksums of the reference output.
    B, c, h, w, G, r = 2, 32, 112, 112, 64, 4
    f0 = torch.from_numpy(synth.lattice_normalish((B, c, G, G), 11))
    f1 = torch.from_numpy(synth.lattice_normalish((B, c, h, w), 12))
    flow_np = synth.homography_flow(B, G, 13)
    flow_np[1] *= np.float32(1.1)
    flow = torch.from_numpy(flow_np)
    K = (2 * r + 1) ** 2
    idx = torch.stack((torch.randint(0, B, (512,), generator=g), torch.randint(0, K, (512,), generator=g),
                       torch.randint(0, G, (512,), generator=g), torch.randint(0, G, (512,), generator=g)), 1)
    out = local_correlation((B, c, h, w), f0, f1, local_radius=r, num_grid=G, flow=flow)
    probes = out[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]]
    save("g1b_local_corr_scale4", seeds=np.array([11, 12, 13]), probe_idx=t2n(idx), probe_val=t2n(probes),
         sum_per_k=t2n(out.double().sum(dim=(0, 2, 3))), abs_sum=np.float64(out.double().abs().sum().item()),
         out_b0_k40=t2n(out[0, 40]), out_b1_k0=t2n(out[1, 0]),
         shape=np.array([B, c, h, w, G, r]))

    # (c) option coverage: grid_based_correlation=True, num_level=2, flow=None (G == h == w)
    B, c, h, w, G, r = 1, 4, 12, 12, 12, 1
    f0 = torch.randn(B, c, G, G, generator=g)
    f1 = torch.randn(B, c, h, w, generator=g)
    flow = torch.rand(B, 2, G, G, generator=g) * 2.2 - 1.1
    o_grid = local_correlation((B, c, h, w), f0, f1, local_radius=r, num_grid=G, flow=flow, grid_based_correlation=True)
    o_lvl2 = local_correlation((B, c, h, w), f0, f1, local_radius=r, num_grid=G, flow=flow, num_level=2)
    o_none = local_correlation((B, c, h, w), f0, f1, local_radius=r, num_grid=G, flow=None)
    # non-square grid-based window uses num_grid for both axes; cover h != w there too
    f1r = torch.randn(B, c, 10, 16, generator=g)
    o_grid_rect = local_correlation((B, c, 10, 16), f0, f1r, local_radius=2, num_grid=G, flow=flow, grid_based_correlation=True)
    save("g1c_local_corr_options", f0=t2n(f0), f1=t2n(f1), f1_rect=t2n(f1r), flow=t2n(flow), out_grid_based=t2n(o_grid),
         out_num_level2=t2n(o_lvl2), out_flow_none=t2n(o_none), out_grid_based_rect=t2n(o_grid_rect),
         r=np.int64(r), G=np.int64(G))

    # (d) every radius the configs use on a c16 map (exercises each specialised kernel), incl. r=0
    B, c, h, w, G = 1, 16, 24, 24, 8
    f0 = torch.randn(B, c, G, G, generator=g)
    f1 = torch.randn(B, c, h, w, generator=g)
    flow = torch.rand(B, 2, G, G, generator=g) * 2.0 - 1.0
    outs = {f"out_r{r}": t2n(local_correlation((B, c, h, w), f0, f1, local_radius=r, num_grid=G, flow=flow))
            for r in (0, 1, 2, 3, 4, 5, 6, 7)}
    save("g1d_local_corr_radii", f0=t2n(f0), f1=t2n(f1), flow=t2n(flow), G=np.int64(G), **outs)


def g2_corr_softargmax(GFNet):
    """model/network.py:415-440"""
    g = torch.Generator().manual_seed(202)
    f0 = torch.randn(2, 16, 8, 8, generator=g)
    f1 = torch.randn(2, 16, 8, 8, generator=g)
    vol = GFNet.corr_volume(None, f0, f1)
    flow = GFNet.pos_embed(None, vol)
    # rectangular maps, different sizes for A and B
    f0r = torch.randn(1, 8, 5, 7, generator=g)
    f1r = torch.randn(1, 8, 6, 4, generator=g)
    volr = GFNet.corr_volume(None, f0r, f1r)
    flowr = GFNet.pos_embed(None, volr)
    # production shape, flow only; inputs regenerated from synth (x3 so that the softmax is selective)
    f0p = torch.from_numpy(3 * synth.lattice_normalish((1, 64, 32, 32), 21))
    f1p = torch.from_numpy(3 * synth.lattice_normalish((1, 64, 32, 32), 22))
    flowp = GFNet.pos_embed(None, GFNet.corr_volume(None, f0p, f1p))
    save("g2_corr_softargmax", f0=t2n(f0), f1=t2n(f1), vol=t2n(vol), flow=t2n(flow),
         f0_rect=t2n(f0r), f1_rect=t2n(f1r), vol_rect=t2n(volr), flow_rect=t2n(flowr),
         prod_seeds=np.array([21, 22]), flow_prod=t2n(flowp))


def g3_kde(kde):
    """utils/kde.py:4-13"""
    g = torch.Generator().manual_seed(303)
    out = {}
    for N in (512, 4096):
        # clustered 4-d matches in [-1,1]^4, like sampled warp rows
        centers = torch.rand(8, 4, generator=g) * 2 - 1
        x = centers[torch.randint(0, 8, (N,), generator=g)] + 0.08 * torch.randn(N, 4, generator=g)
        out[f"x_{N}"] = t2n(x)
        out[f"density_{N}_full"] = t2n(kde(x, std=0.1, half=False, down=None))
        out[f"density_{N}_down8"] = t2n(kde(x, std=0.1, half=False, down=8))
        out[f"density_{N}_down1"] = t2n(kde(x, std=0.1, half=False, down=1))
        xd = x.double()
        d2 = ((xd[:, None, :] - xd[None, :, :]) ** 2).sum(-1)
        out[f"density_{N}_exact64"] = t2n(torch.exp(-d2 / (2 * 0.1 ** 2)).sum(-1))
        out[f"density_{N}_half"] = t2n(kde(x, std=0.1, half=True, down=None).float())
    x = torch.rand(300, 4, generator=g)
    out["x_std"] = t2n(x)
    out["density_std0.25"] = t2n(kde(x, std=0.25, half=False, down=3))
    save("g3_kde", **out)


def g4_refiner_prefix(network):
    """model/network.py:533-558 (ConvRefiner.forward up to the concat at :555)"""
    torch.manual_seed(404)
    c, disp, r, G, hs, ws, B = 8, 6, 2, 10, 18, 22, 2
    K = (2 * r + 1) ** 2
    dim = 2 * c + disp + K
    ref = network.ConvRefiner(dim, dim, 3, kernel_size=5, dw=True, hidden_blocks=2, displacement_emb="linear",
                              displacement_emb_dim=disp, local_corr_num=r, corr_in_other=True, amp=True,
                              disable_local_corr_grad=True, bn_momentum=0.01).eval()
    # non-trivial BN statistics
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.3)
            m.running_var.uniform_(0.5, 1.5)
    captured = {}
    ref.block1.register_forward_pre_hook(lambda mod, inp: captured.__setitem__("d", inp[0].detach().clone()))
    x = torch.randn(B, c, hs, ws)
    y = torch.randn(B, c, hs, ws)
    flow = torch.rand(B, 2, G, G) * 2.2 - 1.1
    with torch.no_grad():
        dflow, dcert, lc = ref(G, x, y, flow, scale_factor=1.25)
    arrays = dict(x=t2n(x), y=t2n(y), flow=t2n(flow), d=t2n(captured["d"]), local_corr=t2n(lc),
                  delta_flow=t2n(dflow), delta_cert=t2n(dcert), G=np.int64(G), r=np.int64(r),
                  scale_factor=np.float64(1.25), hidden_blocks=np.int64(2))
    for k, v in ref.state_dict().items():
        arrays["sd." + k] = t2n(v)
    # variant without local correlation (the scale-1 refiner: corr_in_other=False)
    dim1 = 2 * c + disp
    ref1 = network.ConvRefiner(dim1, dim1, 3, kernel_size=5, dw=True, hidden_blocks=2, displacement_emb="linear",
                               displacement_emb_dim=disp, local_corr_num=0, corr_in_other=False, amp=True,
                               disable_local_corr_grad=True, bn_momentum=0.01).eval()
    cap1 = {}
    ref1.block1.register_forward_pre_hook(lambda mod, inp: cap1.__setitem__("d", inp[0].detach().clone()))
    with torch.no_grad():
        dflow1, dcert1, lc1 = ref1(G, x, y, flow, scale_factor=1.0)
    assert lc1 is None
    arrays.update(d_nocorr=t2n(cap1["d"]), delta_flow_nocorr=t2n(dflow1), delta_cert_nocorr=t2n(dcert1))
    for k, v in ref1.state_dict().items():
        arrays["sd1." + k] = t2n(v)
    save("g4_refiner_prefix", **arrays)


def _toy_model(network, feat_ch, disp_dim, radius, num_itr, num_grid, hidden_blocks=1):
    """A fake `self` for GFNet.forward/match: real ConvRefiners (toy widths), real corr_volume/pos_embed."""
    GFNet = network.GFNet
    scales = ["16", "8", "4", "2", "1"]
    refiners = {}
    for i, s in enumerate(scales):
        K = (2 * radius[i] + 1) ** 2 if radius[i] > 0 else 0
        dim = 2 * feat_ch[i] + disp_dim[i] + K
        refiners[s] = network.ConvRefiner(dim, dim, 3, kernel_size=5, dw=True, hidden_blocks=hidden_blocks,
                                          displacement_emb="linear", displacement_emb_dim=disp_dim[i],
                                          local_corr_num=radius[i], corr_in_other=radius[i] > 0, amp=True,
                                          disable_local_corr_grad=True, bn_momentum=0.01).eval()
    me = types.SimpleNamespace(conv_refiner=refiners, num_grid=list(num_grid), num_itr=list(num_itr),
                               radius=list(radius), training=False)
    me.corr_volume = lambda a, b: GFNet.corr_volume(me, a, b)
    me.pos_embed = lambda v: GFNet.pos_embed(me, v)
    return me


def g5_forward_loop(network):
    """model/network.py:203-283 coarse-to-fine loop, normal and upsample mode, symmetric."""
    torch.manual_seed(505)
    GFNet = network.GFNet
    feat_ch = [8, 8, 4, 4, 2]
    disp = [4, 4, 2, 2, 2]
    radius = [2, 2, 1, 1, 0]
    sides = {"16": 4, "8": 7, "4": 14, "2": 28, "1": 56}
    num_grid = [4, 4, 8, 16, 32]
    me = _toy_model(network, feat_ch, disp, radius, [1, 2, 1, 1, 1], num_grid)
    B = 1
    pyr0 = {s: torch.randn(B, feat_ch[i], sides[s], sides[s]) for i, s in enumerate(sides)}
    pyr1 = {s: torch.randn(B, feat_ch[i], sides[s], sides[s]) for i, s in enumerate(sides)}
    me.extract_features = lambda x, upsample=False: (dict(pyr0), dict(pyr1))
    im = torch.zeros(B, 3, 56, 56)
    with torch.no_grad():
        corresps = GFNet.forward(me, {"im_A": im, "im_B": im}, symmetric=True)
    arrays = {}
    for i, s in enumerate(sides):
        arrays[f"pyr0.{s}"] = t2n(pyr0[s])
        arrays[f"pyr1.{s}"] = t2n(pyr1[s])
        for k, v in me.conv_refiner[s].state_dict().items():
            arrays[f"sd.{s}.{k}"] = t2n(v)
        for itr, d in corresps[s].items():
            arrays[f"flow.{s}.{itr}"] = t2n(d["flow"])
            arrays[f"cert.{s}.{itr}"] = t2n(d["certainty"])
    # upsample pass: scales 8..1 only, seeded by the finest correspondences, scale_factor 1.25
    sides_up = {"8": 9, "4": 18, "2": 36, "1": 72}
    up0 = {s: torch.randn(B, feat_ch[1 + i], sides_up[s], sides_up[s]) for i, s in enumerate(sides_up)}
    up1 = {s: torch.randn(B, feat_ch[1 + i], sides_up[s], sides_up[s]) for i, s in enumerate(sides_up)}
    me.extract_features = lambda x, upsample=False: (dict(up0), dict(up1))
    me.num_grid_up = [5, 10, 20, 40]
    me.num_itr_up = [2, 1, 1, 1]
    me_refiners_full = me.conv_refiner
    im_up = torch.zeros(B, 3, 72, 72)
    with torch.no_grad():
        corresps_up = GFNet.forward(me, {"im_A": im_up, "im_B": im_up}, symmetric=True, upsample=True,
                                    scale_factor=1.25, pre_corresps=corresps["1"][1])
    for i, s in enumerate(sides_up):
        arrays[f"up0.{s}"] = t2n(up0[s])
        arrays[f"up1.{s}"] = t2n(up1[s])
        for itr, d in corresps_up[s].items():
            arrays[f"upflow.{s}.{itr}"] = t2n(d["flow"])
            arrays[f"upcert.{s}.{itr}"] = t2n(d["certainty"])
    arrays["feat_ch"] = np.array(feat_ch)
    arrays["disp"] = np.array(disp)
    arrays["radius"] = np.array(radius)
    arrays["num_grid"] = np.array(num_grid)
    arrays["num_itr"] = np.array([1, 2, 1, 1, 1])
    arrays["num_grid_up"] = np.array(me.num_grid_up)
    arrays["num_itr_up"] = np.array(me.num_itr_up)
    save("g5_forward_loop", **arrays)


def g6_match_post(network):
    """model/network.py:326-384 with forward() replaced by canned correspondences."""
    torch.manual_seed(606)
    GFNet = network.GFNet
    network.get_tuple_transform_ops = lambda **kw: (lambda ims: ims)  # image resize/normalise is not on the path
    arrays = {}
    Gc, G1 = 4, 12  # coarse grid; finest grid of the first pass
    for tag, symmetric, upsample, attenuate in (("sym_up_att", True, True, True), ("plain", False, False, False),
                                                ("sym_noup_att", True, False, True), ("up_noatt", False, True, False)):
        nb = 2 if symmetric else 1
        G_last = 16 if upsample else G1  # upsample_res (28,28): int(28/14)=2 -> grids [2,4,8,16]
        first = {"16": {1: {"flow": torch.randn(nb, 2, Gc, Gc), "certainty": torch.randn(nb, 1, Gc, Gc)}},
                 "1": {1: {"flow": torch.randn(nb, 2, G1, G1) * 0.7, "certainty": torch.randn(nb, 1, G1, G1) * 2}}}
        second = {"1": {1: {"flow": torch.randn(nb, 2, 16, 16) * 0.7, "certainty": torch.randn(nb, 1, 16, 16) * 2}}}
        calls = []

        def fake_forward(batch, symmetric=False, **kw):
            calls.append(kw)
            return first if len(calls) == 1 else second

        me = types.SimpleNamespace(h_resized=16, w_resized=16, upsample_res=(28, 28), symmetric=symmetric,
                                   upsample_preds=upsample, attenuate_cert=attenuate,
                                   num_grid=[Gc, Gc, 6, 8, G1], num_itr=[1, 1, 1, 1, 1], radius=[2, 2, 1, 1, 0],
                                   forward=fake_forward, train=lambda flag: None)
        if attenuate and not upsample:
            me.num_grid_up = [1, 2, 4, G1]  # the reference reads num_grid_up here even without upsampling
        im = torch.zeros(1, 3, 16, 16)
        warp, cert = GFNet.match(me, im, im)
        assert len(calls) == (2 if upsample else 1)
        src = second if upsample else first
        arrays[f"{tag}.cert16"] = t2n(first["16"][1]["certainty"])
        arrays[f"{tag}.flow"] = t2n(src["1"][1]["flow"])
        arrays[f"{tag}.cert"] = t2n(src["1"][1]["certainty"])
        arrays[f"{tag}.warp"] = t2n(warp)
        arrays[f"{tag}.certainty"] = t2n(cert)
        arrays[f"{tag}.G"] = np.int64(G_last)
        if upsample:
            arrays[f"{tag}.scale_factor"] = np.float64(calls[1]["scale_factor"])
    save("g6_match_post", **arrays)


def g7_sample(network):
    """model/network.py:385-414 with the CPU generator seeded."""
    GFNet = network.GFNet
    g = torch.Generator().manual_seed(707)
    G = 48
    lin = torch.linspace(-1 + 1 / G, 1 - 1 / G, G)
    gy, gx = torch.meshgrid(lin, lin, indexing="ij")
    grid = torch.stack((gx, gy), -1)
    a2b = torch.stack((0.8 * gx + 0.1 * gy, -0.1 * gx + 0.9 * gy), -1) + 0.01 * torch.randn(G, G, 2, generator=g)
    warp = torch.cat((torch.cat((grid, a2b), -1), torch.cat((a2b, grid), -1)), 1)  # (G, 2G, 4)
    cert = torch.rand(G, 2 * G, generator=g) ** 3
    cert[:, :5] = 0
    arrays = dict(warp=t2n(warp), certainty=t2n(cert))
    for mode in ("threshold_balanced", "threshold", "balanced"):
        me = types.SimpleNamespace(sample_mode=mode, sample_thresh=0.05)
        torch.manual_seed(7070)
        m, c = GFNet.sample(me, warp, cert, num=500)
        arrays[f"{mode}.matches"] = t2n(m)
        arrays[f"{mode}.certainty"] = t2n(c)
    # num larger than the candidate count
    me = types.SimpleNamespace(sample_mode="threshold_balanced", sample_thresh=0.05)
    torch.manual_seed(7071)
    m, c = GFNet.sample(me, warp[:8, :8], cert[:8, :8] + 0.01, num=500)
    arrays["small.matches"] = t2n(m)
    arrays["small.certainty"] = t2n(c)
    save("g7_sample", **arrays)


def g8_estimation(cv2_stub):
    """estimation.py:12-45 (auc, convert_coordinates) and :46-92 (demo_estimation ACE arithmetic)
    with cv2.findHomography replaced by a canned matrix (OpenCV is absent: SURVEY 8c)."""
    import estimation
    from PIL import Image
    import tempfile

    rng = np.random.default_rng(808)
    errors = list(rng.gamma(2.0, 3.0, size=200))
    aucs = estimation.auc(errors, [3, 5, 10, 20])
    errs_few = [0.5, 30.0, 2.0]
    aucs_few = estimation.auc(errs_few, [3, 5, 10, 20])
    a = rng.uniform(-1, 1, size=(50, 2)).astype(np.float32)
    b = rng.uniform(-1, 1, size=(50, 2)).astype(np.float32)
    pa, pb = estimation.convert_coordinates(a, b, 640, 480, 320, 200)
    arrays = dict(errors=np.array(errors), aucs=np.array(aucs), errors_few=np.array(errs_few),
                  aucs_few=np.array(aucs_few), conv_a=a, conv_b=b, conv_pa=pa, conv_pb=pb,
                  conv_sizes=np.array([640, 480, 320, 200]))

    # demo_estimation end to end with a fake model and canned findHomography
    tmp = tempfile.mkdtemp()
    w1, h1, w2, h2 = 64, 48, 40, 56
    Image.fromarray(np.zeros((h1, w1, 3), np.uint8)).save(os.path.join(tmp, "a.png"))
    Image.fromarray(np.zeros((h2, w2, 3), np.uint8)).save(os.path.join(tmp, "b.png"))
    H_gt = np.array([[1.05, 0.02, 3.0], [-0.03, 0.97, -2.0], [1e-4, -2e-4, 1.0]])
    with open(os.path.join(tmp, "h.json"), "w") as f:
        json.dump({"H": H_gt.tolist()}, f)
    matches = torch.from_numpy(rng.uniform(-1, 1, size=(100, 4)).astype(np.float32))
    seen = {}

    class FakeModel:
        def match(self, a, b):
            return matches, torch.ones(100)

        def sample(self, m, c, num):
            return m, c

    cases = {}
    for tag, H_pred in (("near", H_gt + np.array([[1e-3, 0, 0.5], [0, -1e-3, 0.2], [0, 0, 0]])),
                        ("far", np.array([[2.0, 0, 100.0], [0, 2.0, 50.0], [0, 0, 1.0]])),
                        ("none", None)):
        def fake_find(pa, pb, method=None, confidence=None, ransacReprojThreshold=None, _H=H_pred):
            seen["pa"], seen["pb"] = np.array(pa), np.array(pb)
            seen["args"] = (method, confidence, ransacReprojThreshold)
            return _H, None

        cv2_stub.findHomography = fake_find
        ace, runtime = estimation.demo_estimation(FakeModel(), os.path.join(tmp, "a.png"), os.path.join(tmp, "b.png"),
                                                  os.path.join(tmp, "h.json"))
        cases[tag] = ace
        arrays[f"demo.{tag}.H_pred"] = np.zeros((3, 3)) if H_pred is None else H_pred
        arrays[f"demo.{tag}.ace"] = np.float64(ace)
    arrays["demo.H_gt"] = H_gt
    arrays["demo.sizes"] = np.array([w1, h1, w2, h2])
    arrays["demo.matches"] = matches.numpy()
    arrays["demo.pos_a"] = seen["pa"]
    arrays["demo.pos_b"] = seen["pb"]
    arrays["demo.find_args"] = np.array([float(seen["args"][0]), seen["args"][1], seen["args"][2]])
    save("g8_estimation", **arrays)


def g9_resize_normalise():
    """N3: the image transform GFNet.match applies (model/network.py:293-346) through the reference's own
    utils.utils.get_tuple_transform_ops.  torchvision is absent, so its two tensor-branch calls are restated in the stub:
    transforms.Resize(size, mode, antialias=None) on a float tensor = F.interpolate(mode, align_corners=False,
    antialias=False) with the int -> mode table of torchvision (2 = bilinear, 3 = bicubic), transforms.Normalize =
    (x - mean) / std.  Inputs: synth lattices in [0,1]; sizes cover up-/down-scaling and the identity."""
    import utils.utils as ru
    import torchvision.transforms as tvt

    class Resize:
        def __init__(self, size, interpolation=3, antialias=None):
            self.size = size
            self.mode = {2: "bilinear", 3: "bicubic"}[int(getattr(interpolation, "value", interpolation))]

        def __call__(self, im):
            return F.interpolate(im[None], size=self.size, mode=self.mode, align_corners=False, antialias=False)[0]

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean).view(3, 1, 1), torch.tensor(std).view(3, 1, 1)

        def __call__(self, im):
            return (im - self.mean) / self.std

    tvt.Resize, tvt.Normalize = Resize, Normalize
    ru.transforms.Resize, ru.transforms.Normalize = Resize, Normalize
    arrays = {}
    cases = [("down", (61, 83), (28, 42)), ("up", (24, 20), (56, 70)), ("same", (28, 28), (28, 28)), ("mixed", (50, 30), (42, 42))]
    for i, (name, (H, W), size) in enumerate(cases):
        a = (synth.lattice_uniform((3, H, W), 900 + i) * 0.5 + 0.5).astype(np.float32)
        b = (synth.lattice_uniform((3, H, W), 950 + i) * 0.5 + 0.5).astype(np.float32)
        arrays[f"{name}.seed"] = np.array([900 + i, 950 + i, H, W, size[0], size[1]])
        for mode_name, mode in (("bicubic", 3), ("bilinear", 2)):
            ops = ru.get_tuple_transform_ops(resize=size, mode=mode, normalize=True)
            oa, ob = ops((torch.from_numpy(a), torch.from_numpy(b)))
            arrays[f"{name}.{mode_name}.a"] = oa.numpy()
            arrays[f"{name}.{mode_name}.b"] = ob.numpy()
    save("g9_resize_normalise", **arrays)


def main():
    torch.set_num_threads(4)
    cv2_stub = install_stubs()
    sys.path.insert(0, REF)
    if sys.argv[1:] == ["g9"]:  # add one fixture without touching the others
        g9_resize_normalise()
        return
    from utils.local_correlation import local_correlation
    from utils.kde import kde
    import model.network as network

    g1_local_correlation(local_correlation)
    g2_corr_softargmax(network.GFNet)
    g3_kde(kde)
    g4_refiner_prefix(network)
    g5_forward_loop(network)
    g6_match_post(network)
    g7_sample(network)
    g8_estimation(cv2_stub)
    g9_resize_normalise()
    meta = {"torch": torch.__version__, "numpy": np.__version__, "reference": "KN-Zhang/GFNet snapshot 2026-01-28",
            "generator": "tests/golden/make_golden.py", "device": "cpu", "dtype": "float32"}
    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT) if f.endswith(".npz"))
    print(f"total {total/1024:.1f} KiB")


if __name__ == "__main__":
    main()
