"""GPU parity of the remaining hot-path kernels (through the C ABI) against the oracle and the
reference-generated goldens.  Tolerances: 1e-4 abs+rel for correlation/flow tensors (north_star);
H within 1e-3 px corner error of the oracle on identical matches."""
import numpy as np
import pytest
import torch

import oracle
import synth
from conftest import assert_close, load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


# ---- A2/A3 corr_volume + pos_embed ------------------------------------------------------------
def test_g2_golden_volume_flow():
    from gfnet_amd import ops

    g = load_golden("g2_corr_softargmax")
    assert_close(host(ops.corr_softargmax(dev(g["f0"]), dev(g["f1"]))), g["flow"], TOL, "fused flow")
    vol, flow = ops.corr_volume(dev(g["f0"]), dev(g["f1"]), with_flow=True)
    assert_close(host(vol), g["vol"], TOL, "vol")
    assert_close(host(flow), g["flow"], TOL, "flow with vol")
    assert_close(host(ops.corr_volume(dev(g["f0"]), dev(g["f1"]))), g["vol"], TOL, "vol only")
    assert_close(host(ops.pos_embed(dev(g["vol"]))), g["flow"], TOL, "pos_embed")
    # rectangular, A and B of different size, 35 / 24 positions (not multiples of 32)
    assert_close(host(ops.corr_softargmax(dev(g["f0_rect"]), dev(g["f1_rect"]))), g["flow_rect"], TOL, "rect flow")
    assert_close(host(ops.corr_volume(dev(g["f0_rect"]), dev(g["f1_rect"]))), g["vol_rect"], TOL, "rect vol")


def test_g2_golden_production_shape():
    from gfnet_amd import ops

    g = load_golden("g2_corr_softargmax")
    s0, s1 = [int(v) for v in g["prod_seeds"]]
    f0 = 3 * synth.lattice_normalish((1, 64, 32, 32), s0)
    f1 = 3 * synth.lattice_normalish((1, 64, 32, 32), s1)
    assert_close(host(ops.corr_softargmax(dev(f0), dev(f1))), g["flow_prod"], TOL, "flow prod")


@pytest.mark.parametrize("C,H0,H1,B", [(64, 48, 48, 2), (64, 32, 32, 3), (16, 12, 20, 2), (7, 9, 5, 1)])
def test_corr_softargmax_vs_oracle(C, H0, H1, B):
    from gfnet_amd import ops

    f0 = 2 * synth.lattice_normalish((B, C, H0, H0), 71)
    f1 = 2 * synth.lattice_normalish((B, C, H1, H1 + 1), 72)
    assert_close(host(ops.corr_softargmax(dev(f0), dev(f1))), oracle.corr_softargmax(f0, f1), TOL, "flow")


@pytest.mark.parametrize("C,H,W,B,sym,half", [(64, 32, 32, 4, True, False), (64, 48, 48, 2, True, False), (64, 33, 40, 2, False, False),
                                              (48, 32, 32, 2, True, False), (64, 32, 32, 2, True, True), (64, 64, 64, 1, False, False)])
def test_corr_softargmax_split_bf16_with_and_without_workspace(C, H, W, B, sym, half):
    """Round 6: on 33..64-channel maps with 32..64-position rows the products run on the bf16 matrix core instruction with both operands
    split three ways; gfn_corr_softargmax_fwd_ws splits the B-positions' operand once into a caller-owned workspace.  Through the C ABI:
    with a workspace == without one (the same pieces, the same MFMA order: bit for bit), both within the flow tolerance of the oracle
    (measured ~1e-7: closer to float64 than the fp32 chains), symmetric batches against the two one-way calls, fp16 maps against their
    widened copy, a too-small / misaligned / absent workspace is simply not used, and the size query is 0 for shapes that take none."""
    from gfnet_amd import _lib, ops

    L = _lib.lib()
    f0 = 2 * synth.lattice_normalish((B, C, H, W), 171)
    f1 = 2 * synth.lattice_normalish((B, C, H, W), 172)
    a, b = dev(f0), dev(f1)
    if half:
        a, b = a.half(), b.half()
    nb = 2 * B if sym else B
    need = int(L.gfn_corr_softargmax_ws_bytes(nb, C, H, W))
    assert need == nb * H * (2 if W > 32 else 1) * 12 * 64 * 16
    assert int(L.gfn_corr_softargmax_ws_bytes(nb, 16, H, W)) == 0 and int(L.gfn_corr_softargmax_ws_bytes(nb, C, 16, 16)) == 0

    def run(ws, nbytes):
        flow = torch.empty((nb, 2, H, W), device="cuda")
        _lib.check(L.gfn_corr_softargmax_fwd_ws(_lib.ptr(a), _lib.ptr(b), _lib.GFN_F16 if half else _lib.GFN_F32, _lib.ptr(flow), nb, C, H, W, H, W,
                                                1 if sym else 0, _lib.ptr(ws), nbytes, _lib.stream_ptr(a.device)), "gfn_corr_softargmax_fwd_ws")
        return host(flow)

    ws = torch.empty(need + 16, device="cuda", dtype=torch.uint8)
    with_ws = run(ws, need)
    assert np.array_equal(with_ws, run(None, 0))
    assert np.array_equal(with_ws, run(ws, need - 1))                 # too small: not used
    assert np.array_equal(with_ws, run(ws[8:], need))                 # misaligned: not used
    assert np.array_equal(with_ws, host(ops.corr_softargmax(a, b, symmetric=sym)))
    af, bf = host(a.float()), host(b.float())
    ref = oracle.corr_softargmax(af, bf)
    if sym:
        ref = np.concatenate((ref, oracle.corr_softargmax(bf, af)))
    assert_close(with_ws, ref, TOL, "flow")
    if half:
        assert np.array_equal(with_ws, host(ops.corr_softargmax(a.float(), b.float(), symmetric=sym)))


# ---- A8 kde -------------------------------------------------------------------------------------
@pytest.mark.parametrize("N", [512, 4096])
def test_g3_golden_kde(N):
    from gfnet_amd.utils.kde import kde

    g = load_golden("g3_kde")
    x = dev(g[f"x_{N}"])
    np.testing.assert_allclose(host(kde(x, 0.1, half=False, down=None)), g[f"density_{N}_full"], rtol=2e-4)
    np.testing.assert_allclose(host(kde(x, 0.1, half=False, down=8)), g[f"density_{N}_down8"], rtol=2e-4)
    np.testing.assert_allclose(host(kde(x, 0.1, half=False, down=1)), g[f"density_{N}_down1"], rtol=2e-4)
    np.testing.assert_allclose(host(kde(x, 0.1, half=False)), g[f"density_{N}_exact64"], rtol=1e-4)
    h = kde(x, 0.1, half=True)
    assert h.dtype == torch.float16
    np.testing.assert_allclose(host(h.float()), oracle.kde(g[f"x_{N}"], 0.1, half=True), rtol=2e-3)


def test_kde_std_down_odd_sizes_and_batch():
    from gfnet_amd import ops
    from gfnet_amd.utils.kde import kde

    g = load_golden("g3_kde")
    np.testing.assert_allclose(host(kde(dev(g["x_std"]), 0.25, half=False, down=3)), g["density_std0.25"], rtol=2e-4)
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, size=(3, 777, 4)).astype(np.float32)
    out = host(ops.kde_density(dev(x), std=0.15))
    for b in range(3):
        np.testing.assert_allclose(out[b], oracle.kde(x[b], 0.15, half=False), rtol=1e-4)
    x5 = rng.uniform(-1, 1, size=(300, 5)).astype(np.float32)  # generic point dimension
    np.testing.assert_allclose(host(ops.kde_density(dev(x5), std=0.3)), oracle.kde(x5, 0.3, half=False), rtol=1e-4)


def test_kde_full_size_matches_oracle():
    from gfnet_amd.utils.kde import kde

    rng = np.random.default_rng(1)
    centers = rng.uniform(-1, 1, size=(16, 4))
    x = (centers[rng.integers(0, 16, 20000)] + 0.05 * rng.standard_normal((20000, 4))).astype(np.float32)
    got = host(kde(dev(x), 0.1, half=False, down=None))
    np.testing.assert_allclose(got, oracle.kde(x, 0.1, half=False), rtol=1e-4)


# ---- A4 refiner input, grid_sample, interpolate ---------------------------------------------------
def test_g4_golden_refiner_input():
    from gfnet_amd import ops

    g = load_golden("g4_refiner_prefix")
    d = ops.refiner_input(int(g["G"]), dev(g["x"]), dev(g["y"]), dev(g["flow"]), dev(g["sd.disp_emb.weight"]),
                          dev(g["sd.disp_emb.bias"]), int(g["r"]), scale_factor=float(g["scale_factor"]))
    assert_close(host(d), g["d"], TOL, "d")
    d1 = ops.refiner_input(int(g["G"]), dev(g["x"]), dev(g["y"]), dev(g["flow"]), dev(g["sd1.disp_emb.weight"]),
                           dev(g["sd1.disp_emb.bias"]), 0, scale_factor=1.0, corr_in_other=False)
    assert_close(host(d1), g["d_nocorr"], TOL, "d (no corr)")


def test_refiner_input_production_shape_vs_oracle():
    from gfnet_amd import ops

    B, c, hs, G, r, Dd = 2, 32, 112, 64, 4, 32
    x = synth.lattice_normalish((B, c, hs, hs), 81)
    y = synth.lattice_normalish((B, c, hs, hs), 82)
    flow = synth.homography_flow(B, G, 83)
    w = synth.lattice_uniform((Dd, 2, 1, 1), 84)
    bias = synth.lattice_uniform((Dd,), 85)
    d = ops.refiner_input(G, dev(x), dev(y), dev(flow), dev(w), dev(bias), r, scale_factor=1.25)
    assert_close(host(d), oracle.refiner_input(G, x, y, flow, w, bias, r, scale_factor=1.25), TOL, "d")


def test_grid_sample_and_interpolate_vs_oracle():
    from gfnet_amd import ops

    x = synth.lattice_normalish((2, 5, 13, 17), 91)
    grid = 1.3 * synth.lattice_uniform((2, 7, 9, 2), 92)
    assert_close(host(ops.grid_sample(dev(x), dev(grid))), oracle.grid_sample(x, grid), 1e-5, "grid_sample")
    for size in [(5, 5), (26, 34), (13, 17), (40, 3)]:
        assert_close(host(ops.interpolate_bilinear(dev(x), size)), oracle.interpolate_bilinear(x, size), 1e-6,
                     f"interp {size}")
    g5 = load_golden("g5_forward_loop")  # flow upsampling between scales on reference data
    f = g5["flow.2.1"]
    assert_close(host(ops.interpolate_bilinear(dev(f), 32)), oracle.interpolate_bilinear(f, 32), 1e-6, "flow up")
    # flow + certainty in one launch: bit-identical to the two single calls
    c = synth.lattice_normalish((f.shape[0], 1) + f.shape[2:], 93)
    fa, ca = ops.interpolate_bilinear_pair(dev(f), dev(c), 32)
    assert np.array_equal(host(fa), host(ops.interpolate_bilinear(dev(f), 32)))
    assert np.array_equal(host(ca), host(ops.interpolate_bilinear(dev(c), 32)))
    assert_close(host(ca), oracle.interpolate_bilinear(c, 32), 1e-6, "certainty up")
    fb, cb = ops.interpolate_bilinear_pair(dev(x), dev(x[:, :2]), (9, 40))
    assert_close(host(fb), oracle.interpolate_bilinear(x, (9, 40)), 1e-6, "pair a")
    assert_close(host(cb), oracle.interpolate_bilinear(x[:, :2], (9, 40)), 1e-6, "pair b")


# ---- A5 flow update, A6 match post ------------------------------------------------------------------
def test_flow_update_vs_oracle():
    from gfnet_amd import ops

    B, G = 2, 12
    flow = synth.lattice_uniform((B, 2, G, G), 101)
    cert = synth.lattice_uniform((B, 1, G, G), 102)
    delta = 3 * synth.lattice_uniform((B, 3, G, G), 103)
    prev0 = np.full((B, 2, G, G), 1e-7, np.float32)
    f1, c1, d1 = oracle.flow_update(flow, cert, delta[:, :2], delta[:, 2:3], prev0, 8, 448, 448)
    tf, tc, tp = dev(flow), dev(cert), torch.zeros(B, 2, G, G, device="cuda")
    ops.flow_update_(tf, tc, dev(delta), tp, 8, 448, 448, zero_small=True, first_iteration=True)
    assert_close(host(tf), f1, 1e-6, "flow")
    assert_close(host(tc), c1, 1e-6, "cert")
    assert_close(host(tp), d1, 1e-6, "disp")
    # second iteration with an identical delta: the relative change is 0 < 1e-6 -> displacement zeroed
    f2, c2, d2 = oracle.flow_update(f1, c1, delta[:, :2], delta[:, 2:3], d1, 8, 448, 448)
    ops.flow_update_(tf, tc, dev(delta), tp, 8, 448, 448, zero_small=True, first_iteration=False)
    assert np.all(d2 == 0)
    assert_close(host(tp), d2, 1e-6, "disp 2")
    assert_close(host(tf), f2, 1e-6, "flow 2")
    # out-of-place form on channel slices of one refiner output and on two separate tensors: same numbers, inputs untouched
    td = dev(delta)
    tf0, tc0, tp0 = dev(flow), dev(cert), torch.zeros(B, 2, G, G, device="cuda")
    fo, co = ops.flow_update(tf0, tc0, td[:, :2], td[:, 2:3], tp0, 8, 448, 448, zero_small=True, first_iteration=True)
    assert_close(host(fo), f1, 1e-6, "flow (out of place)")
    assert_close(host(co), c1, 1e-6, "cert (out of place)")
    np.testing.assert_array_equal(host(tf0), flow)
    np.testing.assert_array_equal(host(tc0), cert)
    tp1 = torch.zeros(B, 2, G, G, device="cuda")
    fo2, co2 = ops.flow_update(tf0, tc0, td[:, :2].contiguous(), td[:, 2:3].contiguous(), tp1, 8, 448, 448, first_iteration=True)
    assert torch.equal(fo, fo2) and torch.equal(co, co2) and torch.equal(tp0, tp1)
    # a scale with a single iteration carries nothing: disp_prev=None gives the same flow / certainty
    fo3, co3 = ops.flow_update(tf0, tc0, td[:, :2], td[:, 2:3], None, 8, 448, 448, first_iteration=True)
    assert torch.equal(fo, fo3) and torch.equal(co, co3)
    with pytest.raises(ValueError):
        ops.flow_update(tf0, tc0, td[:, :2], td[:, 2:3], None, 8, 448, 448, first_iteration=False)


@pytest.mark.parametrize("tag,symmetric,attenuate", [("sym_up_att", True, True), ("plain", False, False),
                                                       ("sym_noup_att", True, True), ("up_noatt", False, False)])
def test_g6_golden_match_post(tag, symmetric, attenuate):
    from gfnet_amd import ops

    g = load_golden("g6_match_post")
    warp, cert = ops.match_post(dev(g[f"{tag}.flow"]), dev(g[f"{tag}.cert"]),
                                dev(g[f"{tag}.cert16"]) if attenuate else None, symmetric=symmetric)
    assert_close(host(warp)[0], g[f"{tag}.warp"], 1e-6, "warp")
    assert_close(host(cert)[0], g[f"{tag}.certainty"], 2e-6, "certainty")


# ---- A9 homography solve ---------------------------------------------------------------------------
def _ace(Ha, Hb, S=448):
    return oracle.corner_error(Ha, Hb, S, S, clamp=1e9)


def _points(seed, Bt, N, noise, outliers):
    from test_homography_cpu import make_points, random_h

    rng = np.random.default_rng(seed)
    Hs = [random_h(rng) for _ in range(Bt)]
    return Hs, np.stack([make_points(rng, H, N, noise=noise, outliers=outliers) for H in Hs])


def test_convert_matches_bit_exact():
    from gfnet_amd import ops

    g = load_golden("g8_estimation")
    w1, h1, w2, h2 = [int(v) for v in g["demo.sizes"]]
    pts = host(ops.convert_matches(dev(g["demo.matches"]), w1, h1, w2, h2))
    np.testing.assert_array_equal(pts, oracle.convert_matches(g["demo.matches"], w1, h1, w2, h2))
    np.testing.assert_allclose(pts[:, :2], g["demo.pos_a"], rtol=1e-6)
    np.testing.assert_allclose(pts[:, 2:], g["demo.pos_b"], rtol=1e-6)


@pytest.mark.parametrize("stage", [1, 2, 0])
@pytest.mark.parametrize("confidence", [0.0, 0.99999])
def test_ransac_matches_oracle_hypothesis_for_hypothesis(stage, confidence):
    """Both control flows: all hypotheses (confidence 0) and OpenCV's confidence-driven iteration bound (estimation.py:66-72),
    which is sequential in the oracle and chunked on the device -- chosen hypothesis, inlier count, mask and the bound the
    loop stopped at must be bit-identical."""
    from gfnet_amd import ops

    Hs, pts = _points(21, 4, 3000, noise=0.6, outliers=0.35)
    H, ninl, best, mask, used = ops.find_homography(dev(pts), thresh=3.0, iters=512, seed=5, stage=stage, return_mask=True,
                                                    confidence=confidence, return_iters=True)
    Ho, no, bo, mo, uo = oracle.homography_ransac(pts, thresh=3.0, iters=512, seed=5, stage=stage, return_mask=True, confidence=confidence,
                                                  return_iters=True)
    np.testing.assert_array_equal(host(best), bo)   # same RNG, same counts, same tie-breaking
    np.testing.assert_array_equal(host(ninl), no)
    np.testing.assert_array_equal(host(mask), mo)
    np.testing.assert_array_equal(host(used), uo)
    if confidence > 0:
        assert (uo < 512).all() and (uo > 0).all(), uo  # 65 % inliers: ~60 iterations are enough for 1 - 1e-5
    Hg = host(H)
    for b in range(4):
        assert _ace(Ho[b], Hg[b]) < 1e-3, (stage, b, _ace(Ho[b], Hg[b]))  # north_star: within 1e-3 px of the reference path
        if stage == 2:  # DLT null vector: LU + inverse iteration on the GPU, Jacobi in the oracle -- same vector up to the
            # conditioning of the normal matrix (its 1e-16 summation-order differences over an eigen-gap of ~1e-8)
            assert _ace(Ho[b], Hg[b]) < 1e-4, (b, _ace(Ho[b], Hg[b]))
        if stage == 0:
            assert _ace(Hs[b], Hg[b]) < 0.3


@pytest.mark.parametrize("outliers,lo,hi", [(0.0, 0, 12), (0.2, 12, 40), (0.5, 100, 400), (0.8, 2000, 2000)])
def test_ransac_early_termination_follows_the_inlier_ratio(outliers, lo, hi):
    """niters = log(1 - 0.99999) / log(1 - w^4): 12 at w = 1, ~23 at 0.8, ~180 at 0.5, never below maxIters at 0.2; the device
    stops where the oracle stops, over several chunks of 64 hypotheses and for ragged N."""
    from gfnet_amd import ops

    Hs, pts = _points(31, 5, 2977, noise=0.4, outliers=outliers)
    H, ninl, best, used = ops.find_homography(dev(pts), iters=2000, seed=3, return_iters=True)
    Ho, no, bo, uo = oracle.homography_ransac(pts, iters=2000, seed=3, return_iters=True)
    np.testing.assert_array_equal(host(used), uo)
    np.testing.assert_array_equal(host(best), bo)
    np.testing.assert_array_equal(host(ninl), no)
    assert (uo >= lo).all() and (uo <= hi).all(), uo
    assert (bo >= 0).all()  # (the last update may leave the bound below the index of the winner: w = 1 sets it to 0)
    if outliers <= 0.5:
        for b in range(5):
            assert _ace(Ho[b], host(H)[b]) < 1e-3 and _ace(Hs[b], host(H)[b]) < 0.5


def test_ransac_rejects_degenerate_subsets():
    """checkSubset: with most correspondences on one line a fixed sampler keeps drawing collinear minimal sets; the solver
    re-draws them and still finds the homography carried by the off-line points."""
    from gfnet_amd import ops

    rng = np.random.default_rng(41)
    Hs, pts = _points(41, 1, 2000, noise=0.0, outliers=0.0)
    line = pts[0].copy()
    n_line = 1500
    t = rng.uniform(20, 420, n_line)
    line[:n_line, 0], line[:n_line, 1] = t, 0.5 * t + 30.0  # collinear in image A ...
    x = np.concatenate((line[:n_line, :2], np.ones((n_line, 1))), 1) @ Hs[0].T
    line[:n_line, 2:] = (x[:, :2] / x[:, 2:]).astype(np.float32)  # ... and consistent under H
    p = line[None]
    H, ninl, best, used = ops.find_homography(dev(p), iters=2000, seed=2, return_iters=True)
    Ho, no, bo, uo = oracle.homography_ransac(p, iters=2000, seed=2, return_iters=True)
    np.testing.assert_array_equal(host(best), bo)
    np.testing.assert_array_equal(host(used), uo)
    assert host(ninl)[0] == 2000 and _ace(Hs[0], host(H)[0]) < 1e-2


def test_ransac_noise_free_and_failure_convention():
    from gfnet_amd import ops

    Hs, pts = _points(22, 2, 5000, noise=0.0, outliers=0.0)
    H, ninl, best = ops.find_homography(dev(pts), iters=64, seed=1)
    assert list(host(ninl)) == [5000, 5000] and list(host(best)) == [0, 0]
    for b in range(2):
        assert _ace(Hs[b].astype(np.float32), host(H)[b]) < 1e-3
    # degenerate input -> diag(0,0,1), estimation.py:74-76
    H, ninl, best = ops.find_homography(torch.ones(1, 50, 4, device="cuda"), iters=16)
    np.testing.assert_array_equal(host(H)[0], np.diag([0.0, 0.0, 1.0]))
    assert host(best)[0] == -1
    H, ninl, best = ops.find_homography(torch.zeros(1, 3, 4, device="cuda"), iters=16)
    np.testing.assert_array_equal(host(H)[0], np.diag([0.0, 0.0, 1.0]))


def test_weighted_grid_dlt_matches_oracle():
    from gfnet_amd import ops

    Hs, pts = _points(23, 3, 4001, noise=0.4, outliers=0.0)  # odd count: a ragged last round of the per-thread accumulation
    w = np.random.default_rng(3).uniform(0.05, 1.0, size=(3, 4001)).astype(np.float32)
    H, ok = ops.homography_dlt(dev(pts), dev(w))
    Ho, oko = oracle.homography_dlt(pts, w.astype(np.float64))
    assert host(ok).all() and oko.all()
    for b in range(3):
        assert _ace(Ho[b], host(H)[b]) < 1e-4, _ace(Ho[b], host(H)[b])
        assert _ace(Hs[b], host(H)[b]) < 0.2
    H1, _ = ops.homography_dlt(dev(pts), None)
    Ho1, _ = oracle.homography_dlt(pts)
    assert _ace(Ho1[0], host(H1)[0]) < 1e-3


@pytest.mark.parametrize("B,c,hs,G,r,Dd", [(3, 16, 84, 48, 2, 16), (2, 32, 112, 64, 4, 32), (3, 32, 70, 80, 4, 8), (1, 8, 90, 50, 0, 8),
                                            (5, 8, 150, 96, 0, 8), (2, 16, 100, 47, 2, 16)])
def test_banded_block_order_of_symmetric_batches_changes_nothing(B, c, hs, G, r, Dd):
    """Round 4: symmetric batches with >= 8 cell blocks per direction launch refiner_input's blocks in XCD-banded order (a 1-D grid:
    an XCD takes one band of both directions of a pair; bands of unequal length rotate with the pair, spare slots return at once).
    The concatenated batch -- same maps, not symmetric -- takes the plain 2-D launch: every plane of `d`, the local correlation
    behind the fused plan included, is bit-identical.  Block counts 9, 16, 25, 10 (ragged last block), 36, 9; odd pair counts."""
    from gfnet_amd import ops

    a = synth.lattice_normalish((B, c, hs, hs), 311)
    b = synth.lattice_normalish((B, c, hs, hs), 312)
    flow = np.concatenate((synth.homography_flow(B, G, 313), synth.homography_flow(B, G, 314, scale=0.95)))
    w = synth.lattice_uniform((Dd, 2, 1, 1), 315)
    bias = synth.lattice_uniform((Dd,), 316)
    kw = dict(corr_in_other=r > 0)
    d_sym = ops.refiner_input(G, dev(a), dev(b), dev(flow), dev(w), dev(bias), r, **kw)
    d_cat = ops.refiner_input(G, dev(np.concatenate((a, b))), dev(np.concatenate((b, a))), dev(flow), dev(w), dev(bias), r, **kw)
    np.testing.assert_array_equal(host(d_sym), host(d_cat))
    assert_close(host(d_sym), oracle.refiner_input(G, np.concatenate((a, b)), np.concatenate((b, a)), flow, w, bias, r, **kw), TOL, "d")


# ---- symmetric batches without the reference's concatenated pyramid copies (network.py:213-222) ----
def test_symmetric_virtual_batch_equals_concatenated_batch():
    from gfnet_amd import ops

    B, c, hs, G, r, Dd = 2, 32, 56, 32, 4, 8
    a = synth.lattice_normalish((B, c, hs, hs), 111)
    b = synth.lattice_normalish((B, c, hs, hs), 112)
    flow = np.concatenate((synth.homography_flow(B, G, 113), synth.homography_flow(B, G, 114, scale=0.95)))
    w = synth.lattice_uniform((Dd, 2, 1, 1), 115)
    bias = synth.lattice_uniform((Dd,), 116)
    d_sym = ops.refiner_input(G, dev(a), dev(b), dev(flow), dev(w), dev(bias), r)
    d_cat = ops.refiner_input(G, dev(np.concatenate((a, b))), dev(np.concatenate((b, a))), dev(flow), dev(w), dev(bias), r)
    np.testing.assert_array_equal(host(d_sym), host(d_cat))
    assert_close(host(d_sym), oracle.refiner_input(G, np.concatenate((a, b)), np.concatenate((b, a)), flow, w, bias, r), TOL, "d")
    a16 = 2 * synth.lattice_normalish((B, 64, 16, 16), 117)
    b16 = 2 * synth.lattice_normalish((B, 64, 16, 16), 118)
    f_sym = ops.corr_softargmax(dev(a16), dev(b16), symmetric=True)
    f_cat = ops.corr_softargmax(dev(np.concatenate((a16, b16))), dev(np.concatenate((b16, a16))))
    np.testing.assert_array_equal(host(f_sym), host(f_cat))


@pytest.mark.parametrize("Bt,N", [(3, 20000), (2, 1000), (1, 1), (2, 1025)])
def test_morton_sort_is_the_stable_sort_of_the_keys(Bt, N):
    from gfnet_amd import ops
    from gfnet_amd import _lib

    rng = np.random.default_rng(11)
    x = rng.uniform(-1.1, 1.1, size=(Bt, N, 4)).astype(np.float32)
    x[:, : N // 3, :2] = np.round(x[:, : N // 3, :2] * 4) / 4        # many equal keys: stability matters
    xs, perm = ops._morton_sorted(dev(x), torch.device("cuda"))
    keys = torch.empty((Bt, N), device="cuda", dtype=torch.int32)
    _lib.check(_lib.lib().gfn_kde_morton_keys(_lib.ptr(dev(x)), _lib.ptr(keys), Bt * N, _lib.stream_ptr(torch.device("cuda"))), "keys")
    want = torch.sort(keys, dim=1, stable=True)[1]
    assert torch.equal(perm, want)
    np.testing.assert_array_equal(host(xs), np.take_along_axis(x, host(want)[..., None], axis=1))


def test_kde_culled_equals_dense_and_oracle_on_match_like_points():
    """Spatially culled KDE (Morton-sorted blocks, 6.7-std cut-off) vs the dense kernel and the oracle on
    points shaped like sampled warp rows: A positions over the image, B = homography(A) + noise, some outliers."""
    from gfnet_amd import ops

    rng = np.random.default_rng(7)
    Bt, N = 2, 20000
    a = rng.uniform(-1, 1, size=(Bt, N, 2))
    b = np.stack([0.85 * a[..., 0] + 0.1 * a[..., 1] + 0.05, -0.08 * a[..., 0] + 0.9 * a[..., 1] - 0.03], -1)
    b += 0.01 * rng.standard_normal(b.shape)
    b[:, :2000] = rng.uniform(-1, 1, size=(Bt, 2000, 2))  # outliers
    x = np.concatenate((a, b), -1).astype(np.float32)
    dense = host(ops.kde_density(dev(x), std=0.1, cull=False))
    culled = host(ops.kde_density(dev(x), std=0.1, cull=True))
    # the culled path forms the exponents on the matrix core (|x|^2 + |y|^2 - 2 x.y in bf16x3 pieces, fp32 accumulation):
    # ~1e-5 relative on a term, against the difference form of the dense kernel
    np.testing.assert_allclose(culled, dense, rtol=1e-4)
    print("culled vs dense max rel", np.max(np.abs(culled - dense) / dense))
    for bt in range(Bt):
        ref = oracle.kde(x[bt], 0.1, half=False)
        print("culled vs oracle max rel", np.max(np.abs(culled[bt] - ref) / ref), "dense vs oracle", np.max(np.abs(dense[bt] - ref) / ref))
        np.testing.assert_allclose(culled[bt], ref, rtol=1e-4)
    # separate reference set (x[::8], the reference's CPU-branch subsampling)
    y = np.ascontiguousarray(x[:, ::8])
    c2 = host(ops.kde_density(dev(x), dev(y), std=0.1, cull=True))
    np.testing.assert_allclose(c2[0], oracle.kde(x[0], 0.1, half=False, down=8), rtol=1e-4)


@pytest.mark.parametrize("Bt,N", [(1, 4097), (5, 9000), (32, 6000)])
def test_kde_symmetric_path_is_deterministic_and_matches_oracle(Bt, N):
    """queries == points takes the symmetric kernel (upper-triangle blocks; column sums meet in fixed-point
    integer accumulators): bit-identical from run to run whatever order the waves arrive in, ragged last block,
    with and without the split over column blocks (small / large batches)."""
    from gfnet_amd import ops

    rng = np.random.default_rng(Bt * 1000 + N)
    a = rng.uniform(-1, 1, size=(Bt, N, 2))
    x = np.concatenate((a, 0.9 * a + 0.02 * rng.standard_normal(a.shape)), -1).astype(np.float32)
    xd = dev(x)
    first = host(ops.kde_density(xd, std=0.1, cull=True))
    for _ in range(3):
        assert np.array_equal(first, host(ops.kde_density(xd, std=0.1, cull=True)))
    for bt in (0, Bt - 1):
        np.testing.assert_allclose(first[bt], oracle.kde(x[bt], 0.1, half=False), rtol=1e-4)
    # same points handed in as a separate reference set: the full N x M kernel, same densities
    other = host(ops.kde_density(xd, xd.clone(), std=0.1, cull=True))
    np.testing.assert_allclose(other, first, rtol=2e-5)
    # fp16 rounding of the coordinates / of the density on the device == rounding in torch beforehand (GFNet.sample's use)
    r1 = ops.kde_density(xd, std=0.1, cull=True, round_fp16=True)
    r2 = ops.kde_density(xd.half().float(), std=0.1, cull=True)
    np.testing.assert_allclose(host(r1), host(r2), rtol=2e-5)  # same values; the curve keys see unrounded points, so the order of the sums differs
    assert torch.equal(ops.kde_density(xd[:, :500], std=0.1, cull=False, round_fp16=True),
                       ops.kde_density(xd[:, :500].half().float(), std=0.1, cull=False))
    assert torch.equal(ops.balance_weights(r1, round_fp16=True), ops.balance_weights(r1.half().float()))


# ---- N3 image resize + normalise ----------------------------------------------------------------------
@pytest.mark.parametrize("case", ["down", "up", "same", "mixed"])
def test_g9_resize_normalise_hip(case):
    from gfnet_amd import ops

    g = load_golden("g9_resize_normalise")
    sa, sb, H, W, h, w = (int(v) for v in g[f"{case}.seed"])
    a = (synth.lattice_uniform((3, H, W), sa) * 0.5 + 0.5).astype(np.float32)
    b = (synth.lattice_uniform((3, H, W), sb) * 0.5 + 0.5).astype(np.float32)
    x = np.stack((a, b))
    for mode in ("bicubic", "bilinear"):
        got = host(ops.resize_normalise(dev(x), (h, w), mode))
        assert_close(got[0], g[f"{case}.{mode}.a"], 1e-5, f"{case} {mode} a")
        assert_close(got[1], g[f"{case}.{mode}.b"], 1e-5, f"{case} {mode} b")
        assert_close(got, oracle.resize_normalise(x, (h, w), mode), 1e-5, "vs oracle")


def test_resize_normalise_image_sizes_and_extra_channel():
    """448 / 560 targets from an odd-sized RGBA-like tensor (the 4th channel is ignored, utils/utils.py:110-114)."""
    from gfnet_amd import ops
    from gfnet_amd._lib import GfnError

    x = (synth.lattice_uniform((1, 4, 301, 517), 77) * 0.5 + 0.5).astype(np.float32)
    for size, mode in (((448, 448), "bicubic"), ((560, 560), "bilinear")):
        got = host(ops.resize_normalise(dev(x), size, mode))
        assert got.shape == (1, 3) + size
        assert_close(got, oracle.resize_normalise(x, size, mode), 1e-5, f"{size} {mode}")
    with pytest.raises(ValueError):
        ops.resize_normalise(dev(x), (8, 8), "nearest")
    with pytest.raises(ValueError):
        ops.resize_normalise(dev(x[:, :2]), (8, 8))


# ---- A7: weighted sampling without replacement ------------------------------------------------------------
@pytest.mark.parametrize("N", [50000, 50001, 49999])  # 16-byte key loads need N % 4 == 0; the others take the scalar path
def test_sample_without_replacement_is_a_valid_draw(N):
    from gfnet_amd import ops

    rng = np.random.default_rng(4)
    Bt, N, K = 3, N, 7000
    w = rng.uniform(0.0, 1.0, size=(Bt, N)).astype(np.float32)
    w[:, ::7] = 0.0                                   # zero-weight entries must never be drawn while positives remain
    w[0, :100] = 1e6                                  # overwhelming weights are (almost surely) all drawn
    idx = host(ops.sample_without_replacement(dev(w), K, seed=123))
    assert idx.shape == (Bt, K) and idx.dtype == np.int64
    for b in range(Bt):
        assert np.all(np.diff(idx[b]) > 0)            # increasing => unique
        assert idx[b].min() >= 0 and idx[b].max() < N
        assert np.all(w[b, idx[b]] > 0)
    assert np.isin(np.arange(100)[np.arange(100) % 7 != 0], idx[0]).all()
    again = host(ops.sample_without_replacement(dev(w), K, seed=123))
    np.testing.assert_array_equal(idx, again)         # same seed, same draw
    other = host(ops.sample_without_replacement(dev(w), K, seed=124))
    assert (other != idx).any()
    # fewer positive entries than requested: the zeros fill up, in index order, every positive one is in
    w2 = np.zeros((1, 1000), np.float32)
    w2[0, 10:20] = 1.0
    got = host(ops.sample_without_replacement(dev(w2), 15, seed=1))[0]
    assert set(range(10, 20)) <= set(got.tolist()) and len(set(got.tolist())) == 15
    # K == N: everything
    np.testing.assert_array_equal(host(ops.sample_without_replacement(dev(w2), 1000, seed=1))[0], np.arange(1000))


def test_gather_matches_and_on_the_fly_threshold():
    from gfnet_amd import ops

    rng = np.random.default_rng(11)
    Bt, N, K = 3, 5000, 700
    m = dev(rng.standard_normal((Bt, N, 4)).astype(np.float32))
    c = dev(rng.uniform(0, 0.2, size=(Bt, N)).astype(np.float32))
    idx = torch.stack([torch.randperm(N, device="cuda")[:K] for _ in range(Bt)])
    om, oc = ops.gather_matches(m, c, idx)
    assert torch.equal(om, torch.gather(m, 1, idx[..., None].expand(Bt, K, 4)))
    assert torch.equal(oc, torch.gather(c, 1, idx))
    ct = ops.threshold_certainty(c, 0.05)                       # network.py:391-393 as a separate pass ...
    om2, oc2 = ops.gather_matches(m, c, idx, one_above=0.05)    # ... and applied by the gather
    assert torch.equal(om2, om) and torch.equal(oc2, torch.gather(ct, 1, idx))
    a = ops.sample_without_replacement(ct, 900, seed=5)         # the draw sees the same weights either way
    b = ops.sample_without_replacement(c, 900, seed=5, one_above=0.05)
    assert torch.equal(a, b)


def test_sample_without_replacement_follows_the_weights():
    """First-draw law: with K = 1 the index is drawn with probability w_i / sum(w); inclusion frequencies of a K-subset
    must match torch.multinomial's (same exponential-race law) within sampling noise."""
    from gfnet_amd import ops

    w = np.array([0.5, 1.0, 2.0, 4.0, 0.0, 8.0, 0.25, 0.25], np.float32)
    rows = 40000
    W = np.tile(w, (rows, 1))
    one = host(ops.sample_without_replacement(dev(W), 1, seed=7))[:, 0]
    freq = np.bincount(one, minlength=8) / rows
    np.testing.assert_allclose(freq, w / w.sum(), atol=0.01)
    k3 = host(ops.sample_without_replacement(dev(W), 3, seed=8))
    incl = np.bincount(k3.ravel(), minlength=8) / rows
    ref = torch.multinomial(dev(W), 3, replacement=False)
    incl_ref = np.bincount(host(ref).ravel(), minlength=8) / rows
    np.testing.assert_allclose(incl, incl_ref, atol=0.015)


@pytest.mark.parametrize("c,hs,G,r,dtype", [(16, 56, 32, 2, torch.float32), (32, 28, 16, 4, torch.float16), (64, 16, 16, 7, torch.float32),
                                            (8, 64, 32, 0, torch.float32)])
def test_refiner_input_reuse_keeps_grid_feature(c, hs, G, r, dtype):
    """Second refiner iteration at a scale (network.py:257-268: same features, new flow): `reuse=` hands the previous concat
    tensor back, the kernels rewrite x_hat, the displacement embedding and the local correlation and leave the grid_feature
    planes (a function of x and the grid only) alone -- bit-identical to a fresh call, for the planned (lean) and the other
    local-correlation paths, fp32 and fp16 maps, symmetric batches."""
    from gfnet_amd import ops

    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, c, hs, hs, generator=g).cuda().to(dtype)
    y = torch.randn(2, c, hs, hs, generator=g).cuda().to(dtype)
    w, b = torch.randn(6, 2, generator=g).cuda(), torch.randn(6, generator=g).cuda()
    flow1 = (torch.rand(4, 2, G, G, generator=g) * 1.6 - 0.8).cuda()
    flow2 = (flow1 + 0.05 * torch.randn(4, 2, G, G, generator=g).cuda()).clamp(-0.95, 0.95)
    d1 = ops.refiner_input(G, x, y, flow1, w, b, r, scale_factor=1.25, corr_in_other=r > 0)
    want = ops.refiner_input(G, x, y, flow2, w, b, r, scale_factor=1.25, corr_in_other=r > 0)
    got = ops.refiner_input(G, x, y, flow2, w, b, r, scale_factor=1.25, corr_in_other=r > 0, reuse=d1)
    assert got.data_ptr() == d1.data_ptr()
    assert torch.equal(got, want)
    with pytest.raises(ValueError):
        ops.refiner_input(G, x, y, flow2, w, b, r, scale_factor=1.25, corr_in_other=r > 0, reuse=d1[:, :-1].contiguous())

