"""GPU parity: gfnet_amd.utils.local_correlation (HIP, through the C ABI) vs the oracle and the
reference-generated goldens.  Tolerance: |d| <= 1e-4 * max(1,|ref|) (BASELINE.json north_star:
correlation tensors within 1e-4 rel fp32; abs+rel because values cross zero)."""
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


def run(f0, f1, flow, r, G, **kw):
    from gfnet_amd.utils.local_correlation import local_correlation

    B, c, h, w = f1.shape
    out = local_correlation((B, c, h, w), dev(f0), dev(f1), r, G, flow=None if flow is None else dev(flow), **kw)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_library_reports_gfx950():
    import ctypes
    from gfnet_amd import _lib

    torch.zeros(1).cuda()
    buf = ctypes.create_string_buffer(64)
    _lib.check(_lib.lib().gfn_device_arch(buf, 64), "gfn_device_arch")
    assert buf.value.decode().startswith("gfx950"), buf.value


def test_g1a_golden_general_path():
    g = load_golden("g1a_local_corr_small")  # c=8 -> general per-tap kernel
    out = run(g["f0"], g["f1"], g["flow"], int(g["r"]), int(g["G"]))
    assert_close(out, g["out"], TOL, "g1a")


def test_g1b_golden_scale4_fast_path():
    g = load_golden("g1b_local_corr_scale4")
    B, c, h, w, G, r = [int(v) for v in g["shape"]]
    s0, s1, s2 = [int(v) for v in g["seeds"]]
    f0 = synth.lattice_normalish((B, c, G, G), s0)
    f1 = synth.lattice_normalish((B, c, h, w), s1)
    flow = synth.homography_flow(B, G, s2)
    flow[1] *= np.float32(1.1)
    out = run(f0, f1, flow, r, G)
    idx = g["probe_idx"]
    assert_close(out[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]], g["probe_val"], TOL, "probes")
    assert_close(out[0, 40], g["out_b0_k40"], TOL, "plane b0 k40")
    assert_close(out[1, 0], g["out_b1_k0"], TOL, "plane b1 k0")
    np.testing.assert_allclose(out.astype(np.float64).sum(axis=(0, 2, 3)), g["sum_per_k"], rtol=0, atol=5e-2)
    # and the whole tensor against the oracle, both kernel variants
    ref = oracle.local_correlation((B, c, h, w), f0, f1, r, G, flow=flow)
    assert_close(out, ref, TOL, "fast vs oracle")
    assert_close(run(f0, f1, flow, r, G, _variant=1), ref, TOL, "general vs oracle")


def test_g1c_golden_options():
    g = load_golden("g1c_local_corr_options")
    f0, f1, flow = g["f0"], g["f1"], g["flow"]
    r, G = int(g["r"]), int(g["G"])
    assert_close(run(f0, f1, flow, r, G, grid_based_correlation=True), g["out_grid_based"], TOL, "grid_based")
    assert_close(run(f0, f1, flow, r, G, num_level=2), g["out_num_level2"], TOL, "num_level=2")
    assert_close(run(f0, f1, None, r, G), g["out_flow_none"], TOL, "flow=None")
    assert_close(run(f0, g["f1_rect"], flow, 2, G, grid_based_correlation=True), g["out_grid_based_rect"], TOL,
                 "grid_based rect")


@pytest.mark.parametrize("r", range(8))
def test_g1d_golden_every_radius(r):
    g = load_golden("g1d_local_corr_radii")  # c=16: r=1..7 take the tiled kernel, r=0 the general one
    out = run(g["f0"], g["f1"], g["flow"], r, int(g["G"]))
    assert_close(out, g[f"out_r{r}"], TOL, f"r={r}")


# the four production shapes of basic.json at 448 and the three of the 560 upsample pass
SHAPES = [(64, 32, 32, 7), (64, 56, 32, 6), (32, 112, 64, 4), (16, 224, 128, 2),
          (64, 70, 40, 6), (32, 140, 80, 4), (16, 280, 160, 2)]


@pytest.mark.parametrize("c,hs,G,r", SHAPES)
@pytest.mark.parametrize("flow_kind", ["homography", "zoom", "random", "zoom1.2", "zoom1.35"])
def test_production_shapes_vs_oracle(c, hs, G, r, flow_kind):
    B = 2
    f0 = synth.lattice_normalish((B, c, G, G), 31 + r)
    f1 = synth.lattice_normalish((B, c, hs, hs), 32 + r)
    if flow_kind == "homography":
        flow = synth.homography_flow(B, G, 33)
    elif flow_kind == "zoom":  # magnified + partly outside: tiles fall back to per-round staging / per-tap
        flow = synth.homography_flow(B, G, 34, scale=1.7)
    elif flow_kind.startswith("zoom"):  # mild magnification: regions that fit the stage only without its pitch padding
        flow = synth.homography_flow(B, G, 36, scale=float(flow_kind[4:]))
    else:  # uncorrelated flow: no two neighbouring cells share a window
        flow = 1.2 * synth.lattice_uniform((B, 2, G, G), 35)
    out = run(f0, f1, flow, r, G)
    ref = oracle.local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow)
    assert_close(out, ref, TOL, f"c{c} hs{hs} G{G} r{r} {flow_kind}")


@pytest.mark.parametrize("c,hs,G,r", [(32, 112, 64, 4), (16, 224, 128, 2), (32, 140, 80, 4), (16, 280, 160, 2), (16, 60, 40, 3), (16, 30, 24, 1),
                                      (32, 168, 96, 4), (16, 336, 192, 2)])
@pytest.mark.parametrize("flow_kind", ["homography", "zoom", "random", "border", "rot"])
def test_lean_tile_kernel_is_bit_identical_to_round1_kernel(c, hs, G, r, flow_kind):
    """The round-2 lean tile kernel (variant 4, r <= 4) keeps the round-1 kernel's arithmetic (same D accumulation order, same
    epilogue): the two must agree bit for bit, whatever path a tile takes (staged, second launch, per-tap, empty windows).
    Round 3: where the default path (variant 0) runs the D-stage on the matrix core (split-bf16 operands, r >= 5 on 64-channel maps:
    csrc/local_corr_mq.h) it is a different numerics class: within the tolerance of the fp32 FMA kernels and of the oracle."""
    B = 4
    f0 = synth.lattice_normalish((B, c, G, G), 231 + r)
    f1 = synth.lattice_normalish((B, c, hs, hs), 232 + r)
    if flow_kind == "homography":
        flow = synth.homography_flow(B, G, 233)
    elif flow_kind == "zoom":
        flow = synth.homography_flow(B, G, 234, scale=1.7)
    elif flow_kind == "random":
        flow = 1.2 * synth.lattice_uniform((B, 2, G, G), 235)
    elif flow_kind == "border":  # shifted so that a band of windows hangs over / leaves the image on every side
        flow = synth.homography_flow(B, G, 236, scale=1.02)
        flow[0, 0] += np.float32(0.35); flow[1, 0] -= np.float32(0.4); flow[2, 1] += np.float32(0.3); flow[3, 1] -= np.float32(0.45)
    else:  # rotations by 10..40 degrees about the centre
        lin = (np.arange(G, dtype=np.float64) * 2 + 1) / G - 1
        gy, gx = np.meshgrid(lin, lin, indexing="ij")
        flow = np.empty((B, 2, G, G), np.float32)
        for b in range(B):
            a = np.deg2rad(10.0 * (b + 1))
            flow[b, 0] = (np.cos(a) * gx - np.sin(a) * gy) * 0.9
            flow[b, 1] = (np.sin(a) * gx + np.cos(a) * gy) * 0.9
        flow += np.float32(0.002) * synth.lattice_uniform((B, 2, G, G), 237)
    lean = run(f0, f1, flow, r, G, _variant=4)
    old = run(f0, f1, flow, r, G, _variant=2)
    np.testing.assert_array_equal(lean, old)
    default = run(f0, f1, flow, r, G)
    assert_close(default, lean, TOL, f"default path vs fp32 FMA kernels, {flow_kind}")
    if flow_kind in ("border", "rot") or not np.array_equal(default, lean):
        assert_close(default, oracle.local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow), TOL, f"{flow_kind} vs oracle")


def _flows_of_kind(flow_kind, B, G, seed):
    if flow_kind == "homography":
        return synth.homography_flow(B, G, seed)
    if flow_kind == "zoom":
        return synth.homography_flow(B, G, seed + 1, scale=1.7)
    if flow_kind == "random":
        return 1.2 * synth.lattice_uniform((B, 2, G, G), seed + 2)
    if flow_kind == "noisy":  # smooth flow + a few pixels of noise: groups of 2 x 8 cells whose windows spread past the accumulators
        return synth.homography_flow(B, G, seed + 5) + np.float32(0.18) * synth.lattice_uniform((B, 2, G, G), seed + 6)
    if flow_kind == "border":  # shifted so that a band of windows hangs over / leaves the image on every side
        flow = synth.homography_flow(B, G, seed + 3, scale=1.02)
        flow[0, 0] += np.float32(0.35); flow[1, 0] -= np.float32(0.4); flow[2, 1] += np.float32(0.3); flow[3, 1] -= np.float32(0.45)
        return flow
    lin = (np.arange(G, dtype=np.float64) * 2 + 1) / G - 1  # rotations by 10..40 degrees about the centre
    gy, gx = np.meshgrid(lin, lin, indexing="ij")
    flow = np.empty((B, 2, G, G), np.float32)
    for b in range(B):
        a = np.deg2rad(10.0 * (b + 1))
        flow[b, 0] = (np.cos(a) * gx - np.sin(a) * gy) * 0.9
        flow[b, 1] = (np.sin(a) * gx + np.cos(a) * gy) * 0.9
    return flow + np.float32(0.002) * synth.lattice_uniform((B, 2, G, G), seed + 4)


@pytest.mark.parametrize("hs,G,r", [(32, 32, 7), (56, 32, 6), (70, 40, 6), (48, 48, 7), (45, 27, 5), (37, 21, 6)])
@pytest.mark.parametrize("flow_kind", ["homography", "zoom", "random", "noisy", "border", "rot"])
@pytest.mark.parametrize("f16", [False, True])
def test_large_windows_on_the_matrix_core_vs_fp32_kernel_and_oracle(hs, G, r, flow_kind, f16):
    """r >= 5 on 64-channel maps: the default path multiplies on the matrix core (csrc/local_corr_mq.h: split-bf16 operands, windows
    clipped to the image, tiles that do not fit its accumulators handed to the round-1 routine inside the launch or to the second
    launch).  Every route against the round-1 fp32 FMA kernel (variant 2) and the oracle, fp32 and fp16 maps (odd map sides keep
    fp16 maps on the round-1 kernel), grids that are not multiples of the tile."""
    B, c = 4, 64
    f0 = synth.lattice_normalish((B, c, G, G), 331 + r)
    f1 = synth.lattice_normalish((B, c, hs, hs), 332 + r)
    if f16:
        f1 = (f1 * np.float32(1.37)).astype(np.float16)
    flow = _flows_of_kind(flow_kind, B, G, 340).astype(np.float32)
    default = run(f0, f1, flow, r, G)
    fp32_kernel = run(f0, f1, flow, r, G, _variant=2)
    assert_close(default, fp32_kernel, TOL, f"matrix core vs fp32 FMA kernel, {flow_kind}")
    assert_close(default, oracle.local_correlation((B, c, hs, hs), f0, f1.astype(np.float32), r, G, flow=flow), TOL, f"{flow_kind} vs oracle")


def test_large_windows_edge_cases_rect_maps_tiny_grids_slices_and_wild_flows():
    """The r >= 5 matrix-core path off the beaten track: rectangular maps and grids that are no multiple of the f0 quads (G = 5, 13, 22),
    a single direction, f0 read from / out written into a concat slice (batch strides), non-finite and far-away flows (memory safety,
    zeros padding)."""
    from gfnet_amd.utils.local_correlation import local_correlation

    c = 64
    for (B, h, w, G, r, seed) in [(1, 23, 41, 13, 6, 701), (3, 40, 30, 5, 7, 702), (2, 38, 38, 22, 5, 703)]:
        f0 = synth.lattice_normalish((B, c, G, G), seed)
        f1 = synth.lattice_normalish((B, c, h, w), seed + 1)
        flow = synth.homography_flow(B, G, seed + 2, scale=1.05)
        assert_close(run(f0, f1, flow, r, G), oracle.local_correlation((B, c, h, w), f0, f1, r, G, flow=flow), TOL, f"rect {h}x{w} G{G} r{r}")
    # concat slice in, concat slice out
    B, hs, G, r = 2, 56, 32, 6
    K = (2 * r + 1) ** 2
    f0 = synth.lattice_normalish((B, c, G, G), 711)
    f1 = synth.lattice_normalish((B, c, hs, hs), 712)
    flow = synth.homography_flow(B, G, 713)
    d = torch.full((B, c + 7 + K, G, G), 3.0, device="cuda")
    d[:, :c] = dev(f0)
    local_correlation((B, c, hs, hs), d[:, :c], dev(f1), r, G, flow=dev(flow), out=d[:, c + 7:])
    torch.cuda.synchronize()
    assert_close(d[:, c + 7:].cpu().numpy(), oracle.local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow), TOL, "slice r6")
    assert torch.all(d[:, c:c + 7] == 3.0)
    # wild flows
    flow = synth.homography_flow(B, G, 714)
    flow[0, 0, 0, 0] = 1e30
    flow[0, 1, 3, 3] = -1e9
    flow[0, 0, 5, 5] = 50.0
    flow[1, :, 7, 9] = np.nan
    out = run(f0, f1, flow, r, G)
    ref = oracle.local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow)
    good = np.ones((B, G, G), bool)
    good[0, 0, 0] = good[0, 3, 3] = good[1, 7, 9] = False  # coordinates beyond float->int range / nan: value unspecified, must not crash
    for b in range(B):
        assert_close(out[b][:, good[b]], ref[b][:, good[b]], TOL, "finite cells")
    assert np.all(out[0][:, 5, 5] == 0)  # far outside the image: zeros padding


def test_ragged_sizes_and_rect_maps():
    # G not a multiple of the tile, rectangular f1, batch of 3
    B, c, h, w, G, r = 3, 16, 37, 53, 21, 3
    f0 = synth.lattice_normalish((B, c, G, G), 41)
    f1 = synth.lattice_normalish((B, c, h, w), 42)
    flow = synth.homography_flow(B, G, 43, scale=1.05)
    assert_close(run(f0, f1, flow, r, G), oracle.local_correlation((B, c, h, w), f0, f1, r, G, flow=flow), TOL, "ragged")


def test_non_finite_and_far_flow_is_memory_safe():
    B, c, hs, G, r = 1, 16, 24, 16, 2
    f0 = synth.lattice_normalish((B, c, G, G), 51)
    f1 = synth.lattice_normalish((B, c, hs, hs), 52)
    flow = synth.homography_flow(B, G, 53)
    flow[0, 0, 0, 0] = 1e30
    flow[0, 1, 3, 3] = -1e9
    flow[0, 0, 5, 5] = 50.0
    out = run(f0, f1, flow, r, G)
    ref = oracle.local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow)
    good = np.ones((G, G), bool)
    good[0, 0] = good[3, 3] = False  # coordinates beyond float->int range: value unspecified, must not crash
    assert_close(out[0][:, good], ref[0][:, good], TOL, "finite cells")
    assert np.all(out[0][:, 5, 5] == 0)  # far outside the image: zeros padding


def test_writes_into_concat_slice_and_reads_f0_from_it():
    B, c, hs, G, r = 2, 16, 40, 24, 2
    K = (2 * r + 1) ** 2
    f0 = synth.lattice_normalish((B, c, G, G), 61)
    f1 = synth.lattice_normalish((B, c, hs, hs), 62)
    flow = synth.homography_flow(B, G, 63)
    from gfnet_amd.utils.local_correlation import local_correlation

    d = torch.full((B, c + 5 + K, G, G), 7.0, device="cuda")
    d[:, :c] = dev(f0)
    local_correlation((B, c, hs, hs), d[:, :c], dev(f1), r, G, flow=dev(flow), out=d[:, c + 5:])
    torch.cuda.synchronize()
    ref = oracle.local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow)
    assert_close(d[:, c + 5:].cpu().numpy(), ref, TOL, "slice")
    assert torch.all(d[:, c:c + 5] == 7.0)


def test_errors_are_python_exceptions():
    from gfnet_amd.utils.local_correlation import local_correlation

    f0 = torch.zeros(1, 16, 4, 4, device="cuda")
    f1 = torch.zeros(1, 16, 8, 8, device="cuda")
    with pytest.raises(ValueError):
        local_correlation((1, 16, 8, 8), f0, f1, 2, 4, flow=None)  # flow=None needs G == h == w
    with pytest.raises(ValueError):
        local_correlation((1, 16, 8, 8), f0, f1, 2, 5, flow=torch.zeros(1, 2, 5, 5, device="cuda"))
    with pytest.raises(RuntimeError):
        local_correlation((1, 16, 8, 8), f0.cpu(), f1.cpu(), 2, 4, flow=torch.zeros(1, 2, 4, 4))


# BASELINE config 3: googlemap 672x672 (pyramid sides 48/84/168/336, grids 48/48/96/192) -- larger maps,
# grids that are not multiples of the tile, second-level (sub-tile) staging under strong zoom
@pytest.mark.parametrize("c,hs,G,r", [(64, 48, 48, 7), (64, 84, 48, 6), (32, 168, 96, 4), (16, 336, 192, 2)])
def test_config3_672_shapes_vs_oracle(c, hs, G, r):
    B = 2
    f0 = synth.lattice_normalish((B, c, G, G), 131 + r)
    f1 = synth.lattice_normalish((B, c, hs, hs), 132 + r)
    flow = synth.homography_flow(B, G, 133, scale=1.0)
    flow[1] = synth.homography_flow(1, G, 134, scale=1.45)[0] * np.float32(0.8)  # strong zoom: sub-tile / gather paths
    out = run(f0, f1, flow, r, G)
    ref = oracle.local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow)
    assert_close(out, ref, TOL, f"672: c{c} hs{hs} G{G} r{r}")


# ---- N4: gradient w.r.t. feature0 ---------------------------------------------------------------------
@pytest.mark.parametrize("B,c,h,w,G,r,num_level,grid_based", [(2, 4, 7, 9, 5, 1, 1, False), (1, 19, 12, 12, 8, 2, 1, False),
                                                              (2, 16, 16, 16, 16, 3, 2, False), (1, 8, 10, 14, 6, 2, 1, True)])
def test_local_correlation_backward_feature0(B, c, h, w, G, r, num_level, grid_based):
    """The reference lets gradients reach feature0 only (local_correlation.py:54-60).  autograd through the wrapper
    (gfn_local_corr_bwd_f0) against the oracle's gradient, which is read off the pinned forward by linearity."""
    from gfnet_amd.utils.local_correlation import local_correlation

    f0 = synth.lattice_normalish((B, c, G, G), 501)
    f1 = synth.lattice_normalish((B, c, h, w), 502)
    flow = (synth.lattice_uniform((B, 2, G, G), 503) * 0.9).astype(np.float32)
    K = (2 * r + 1) ** 2 * num_level
    g = synth.lattice_normalish((B, K, G, G), 504)
    t0 = torch.from_numpy(f0).cuda().requires_grad_(True)
    t1 = torch.from_numpy(f1).cuda().requires_grad_(True)
    tf = torch.from_numpy(flow).cuda().requires_grad_(True)
    out = local_correlation((B, c, h, w), t0, t1, r, G, flow=tf, grid_based_correlation=grid_based, num_level=num_level)
    assert out.requires_grad
    want_fwd = oracle.local_correlation((B, c, h, w), f0, f1, r, G, flow=flow, grid_based_correlation=grid_based, num_level=num_level)
    assert_close(out.detach().cpu().numpy(), want_fwd, 1e-4, "forward under autograd")
    (out * torch.from_numpy(g).cuda()).sum().backward()
    assert t1.grad is None and tf.grad is None  # sampling runs under no_grad in the reference
    want = oracle.local_correlation_grad_feature0((B, c, h, w), g, f1, r, G, flow=flow, grid_based_correlation=grid_based,
                                                  num_level=num_level)
    assert_close(t0.grad.cpu().numpy(), want, 1e-4, "grad feature0")


def test_local_correlation_no_grad_paths_unchanged():
    from gfnet_amd.utils.local_correlation import local_correlation

    B, c, h, w, G, r = 1, 16, 12, 12, 8, 2
    t0 = torch.from_numpy(synth.lattice_normalish((B, c, G, G), 511)).cuda().requires_grad_(True)
    t1 = torch.from_numpy(synth.lattice_normalish((B, c, h, w), 512)).cuda()
    fl = torch.from_numpy((synth.lattice_uniform((B, 2, G, G), 513) * 0.9).astype(np.float32)).cuda()
    with torch.no_grad():
        a = local_correlation((B, c, h, w), t0, t1, r, G, flow=fl)
    assert not a.requires_grad
    b = local_correlation((B, c, h, w), t0, t1, r, G, flow=fl)
    assert b.requires_grad and torch.equal(a, b.detach())


def test_local_correlation_fp16_features():
    """fp16 feature storage (BASELINE config 5): inputs are widened to fp32 for the kernels, the result comes back in
    feature0's dtype like the reference's (local_correlation.py returns feature0.dtype math under autocast)."""
    from gfnet_amd.utils.local_correlation import local_correlation

    B, c, h, w, G, r = 2, 32, 28, 28, 16, 4
    f0 = synth.lattice_normalish((B, c, G, G), 601).astype(np.float16)
    f1 = synth.lattice_normalish((B, c, h, w), 602).astype(np.float16)
    flow = (synth.lattice_uniform((B, 2, G, G), 603) * 0.9).astype(np.float32)
    out = local_correlation((B, c, h, w), torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda(), r, G,
                            flow=torch.from_numpy(flow).cuda())
    assert out.dtype == torch.float16
    want = oracle.local_correlation((B, c, h, w), f0.astype(np.float32), f1.astype(np.float32), r, G, flow=flow)
    assert_close(out.float().cpu().numpy(), want, 2e-3, "fp16 features")  # one fp16 rounding of the result


@pytest.mark.parametrize("hs,G,r", [(56, 32, 6), (32, 32, 7)])
def test_large_windows_error_follows_the_operands_not_the_result_and_the_fp32_switch(hs, G, r):
    """the matrix-core path of r >= 5 splits operands into bf16 pairs, so a value is within 2^-17 * sum_c |f0_c f1_c| / sqrt(C)
    of the fp32 result -- a bound in the operands' magnitude.  Features of magnitude ~100 whose channel sums cancel (f0 = a common
    pairs of opposite sign against nearly equal f1 channels) make |result| << sum |products|: the 1e-4 * max(1, |ref|) rule of the unit-scale tests does
    not apply there, the operand bound does, and ops.LOCAL_CORR_FP32 (C-ABI variant 4) gives the fp32 FMA kernel's bits."""
    from gfnet_amd import ops

    B, c = 2, 64
    f0 = synth.lattice_normalish((B, c, G, G), 951 + r) * np.float32(100.0)
    f0[:, 1::2] = -f0[:, 0::2]                            # channel pairs of opposite sign ...
    f1 = synth.lattice_normalish((B, c, hs, hs), 952 + r) * np.float32(100.0)
    f1[:, 1::2] = f1[:, 0::2] * np.float32(1.0 + 1e-3)   # ... against nearly equal ones: every pair's two products cancel to 1e-3
    flow = synth.homography_flow(B, G, 953 + r)
    ref = oracle.local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow)
    mag = oracle.local_correlation((B, c, hs, hs), np.abs(f0), np.abs(f1), r, G, flow=flow)   # sum_c |f0_c| * bilinear(|f1_c|) / sqrt(C)
    assert np.abs(ref).max() < 0.05 * mag.max()          # the sums do cancel
    default = run(f0, f1, flow, r, G)
    fp32 = run(f0, f1, flow, r, G, _variant=2)
    # the fp32 FMA kernel against the float64-accumulating oracle: a few ulp of the partial sums
    assert np.all(np.abs(fp32 - ref) <= 2.0 ** -19 * mag + 1e-6)
    # the split-bf16 products: 2^-17 of the operand magnitude (plus the fp32 summation error above)
    assert np.all(np.abs(default - ref) <= 2.0 ** -16 * mag + 1e-6)
    assert not np.array_equal(default, fp32)
    ops.LOCAL_CORR_FP32 = True
    try:
        switched = run(f0, f1, flow, r, G)
    finally:
        ops.LOCAL_CORR_FP32 = False
    np.testing.assert_array_equal(switched, fp32)
