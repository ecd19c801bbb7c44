"""GPU: the refiner conv stack (csrc/conv_stack.hip) against torch's convs of the same
layers (model/network.py:471-487, 560-563) through the C ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def _maxerr(a, b):
    return float((a.double() - b.double()).abs().max()), float(b.double().abs().max())


def _block_case(B, C, M, G, bias=True):
    x = _rand(B, C, G, G, seed=1)
    w = _rand(C, 1, 5, 5, seed=2, scale=0.3)
    cb = _rand(C, seed=3) if bias else None
    gamma, beta_bn = _rand(C, seed=4).abs() + 0.5, _rand(C, seed=5)
    mean, var = _rand(C, seed=6), _rand(C, seed=7).abs() + 0.1
    pw = _rand(M, C, seed=8, scale=C ** -0.5)
    pb = _rand(M, seed=9)
    eps = 1e-5
    t = F.relu(F.batch_norm(F.conv2d(x.double(), w.double(), cb.double() if bias else None, padding=2, groups=C), mean.double(),
                            var.double(), gamma.double(), beta_bn.double(), False, 0.0, eps))
    want = F.conv2d(t, pw.double().reshape(M, C, 1, 1), pb.double())
    alpha = (gamma.double() / torch.sqrt(var.double() + eps)).float()
    beta = (beta_bn.double() - mean.double() * alpha.double()).float()
    return x, (w, cb, alpha, beta, pw, pb), want


# (B, C, M, G): the five refiner widths, every tile shape (G % 32 == 0 -> 4x32 cells, G % 16 -> 8x16,
# else 16x8), ragged channel counts, partial edge tiles, G not a multiple of 4 (two-pass form)
CASES = [(2, 24, 24, 64), (1, 73, 73, 40), (1, 417, 417, 32), (1, 361, 361, 16), (1, 177, 177, 80), (3, 5, 9, 8), (2, 8, 8, 4),
         (1, 33, 70, 12), (1, 24, 24, 160), (2, 16, 16, 20), (1, 225, 100, 24), (2, 22, 22, 10), (1, 7, 7, 5), (1, 40, 33, 48),
         (1, 73, 73, 128), (2, 24, 24, 80), (1, 90, 96, 32)]


@pytest.mark.parametrize("B,C,M,G", CASES)
def test_conv_block_matches_torch_float64(B, C, M, G):
    from gfnet_amd import ops

    x, (w, cb, alpha, beta, pw, pb), want = _block_case(B, C, M, G, bias=(C % 2 == 1))
    packed = ops.conv_block_pack(w, cb, alpha, beta, pw, pb)
    got = ops.conv_block(x, packed, M)
    assert got.shape == want.shape
    err, mag = _maxerr(got, want)
    assert err <= 1e-5 * max(mag, 1.0), (err, mag)  # fp32 fma chains (25 taps, then K products) vs float64
    two_pass = ops.conv_block(x, packed, M, variant=1)
    assert torch.equal(got, two_pass), "fused and two-pass kernels must agree bit for bit"


@pytest.mark.parametrize("B,C,M,G", CASES)
def test_conv_block_fp16_operands(B, C, M, G):
    """variant 2: W and relu output rounded to fp16, fp32 accumulation.  Checked against float64 math on
    the same rounded operands (tight) and against the unrounded result (fp16 operand error bound)."""
    from gfnet_amd import ops

    x, (w, cb, alpha, beta, pw, pb), want = _block_case(B, C, M, G, bias=(C % 2 == 1))
    packed = ops.conv_block_pack(w, cb, alpha, beta, pw, pb)
    got = ops.conv_block(x, packed, M, variant=2)
    assert torch.equal(got, ops.conv_block(x, packed, M, variant=3)), "fused and two-pass fp16 kernels must agree bit for bit"
    t = F.relu((F.conv2d(x.double(), w.double(), cb.double() if cb is not None else None, padding=2, groups=C)
                * alpha.double().view(1, C, 1, 1) + beta.double().view(1, C, 1, 1)))
    rounded = F.conv2d(t.float().half().double(), pw.half().double().reshape(M, C, 1, 1), pb.double())
    err, mag = _maxerr(got, rounded)
    assert err <= 1e-3 * max(mag, 1.0), (err, mag)  # a few relu outputs round the other way (fp32 vs float64 depthwise)
    err, mag = _maxerr(got, want)
    assert err <= 4e-3 * max(mag, 1.0), (err, mag)


def _to_half_map(x):
    """(B, C, G, G) float -> the half map layout (B, ceil(C/2), G, G, 2) float16, odd channel past C zero."""
    B, C, G, _ = x.shape
    xp = torch.zeros(B, 2 * ((C + 1) // 2), G, G, device=x.device, dtype=torch.float16)
    xp[:, :C] = x.half()
    return xp.reshape(B, (C + 1) // 2, 2, G, G).permute(0, 1, 3, 4, 2).contiguous()


HALF_CASES = [c for c in CASES if c[3] % 4 == 0]


@pytest.mark.parametrize("B,C,M,G", HALF_CASES)
def test_conv_block_half_maps(B, C, M, G):
    """fp16 maps in HBM (the reference's autocast class, network.py:560-562): input and depthwise taps rounded to fp16 (what
    autocast feeds the depthwise conv), fp32 accumulation and BatchNorm, relu output and 1x1 weights rounded to fp16, fp32
    accumulation, the output rounded to fp16 where it is a map.  Checked against float64 math on the same rounded operands."""
    from gfnet_amd import ops

    x, (w, cb, alpha, beta, pw, pb), _ = _block_case(B, C, M, G, bias=(C % 2 == 1))
    packed = ops.conv_block_pack(w, cb, alpha, beta, pw, pb)
    t = F.relu((F.conv2d(x.half().double(), w.half().double(), cb.double() if cb is not None else None, padding=2, groups=C)
                * alpha.double().view(1, C, 1, 1) + beta.double().view(1, C, 1, 1)))
    want = F.conv2d(t.float().half().double(), pw.half().double().reshape(M, C, 1, 1), pb.double())
    mag = max(float(want.abs().max()), 1.0)
    tol = 2e-3 * mag  # fp16 rounding of the output (2^-11 relative) + the few relu outputs that round the other way

    first = ops.conv_block_half(x, packed, C, M)  # fp32 in (rounded by the kernel), half out
    assert first.shape == (B, (M + 1) // 2, G, G, 2) and first.dtype == torch.float16
    assert _maxerr(ops.half_map_to_float(first, M), want)[0] <= tol
    hm = _to_half_map(x)
    mid = ops.conv_block_half(hm, packed, C, M)
    assert torch.equal(mid, first), "fp32 input is rounded to the same half map on the way in"
    if M % 2:  # the odd channel past M is written as zero: the next block reads it
        assert float(mid.reshape(B, (M + 1) // 2, G * G, 2)[:, -1, :, 1].abs().max()) == 0.0
    last = ops.conv_block_half(hm, packed, C, M, out_half=False)
    assert last.dtype == torch.float32 and _maxerr(last, want)[0] <= 0.5 * tol
    assert torch.equal(last.half(), ops.half_map_to_float(mid, M).half())
    with pytest.raises(ValueError):
        ops.conv_block_half(hm, packed, C + 2, M)  # not the map's channel count


def test_conv_block_half_rejects_bad_arguments():
    from gfnet_amd import _lib, ops
    from gfnet_amd._lib import GfnError, ptr, stream_ptr

    x, (w, cb, alpha, beta, pw, pb), _ = _block_case(1, 8, 8, 10)  # G % 4 != 0
    packed = ops.conv_block_pack(w, cb, alpha, beta, pw, pb)
    with pytest.raises(GfnError):
        ops.conv_block_half(x, packed, 8, 8)
    x8 = _rand(1, 8, 8, 8)
    out = torch.empty(1, 8, 8, 8, device="cuda")
    rc = _lib.lib().gfn_conv_block_half_fwd(ptr(x8), _lib.GFN_F32, ptr(packed), ptr(out), _lib.GFN_F32, 1, 8, 8, 8, stream_ptr(x8.device))
    assert rc != 0  # fp32 in and out is gfn_conv_block_fwd's job


@pytest.mark.parametrize("variant", [0, 2])
def test_conv_block_large_grid_two_items_per_workgroup(variant):
    """>= 16384 work items: the launcher gives every workgroup two consecutive tiles (pipelined back to back, zero
    padding refreshed between them); the two-pass kernels are the bit-exact reference."""
    from gfnet_amd import ops

    B, C, G = 64, 8, 256
    x, (w, cb, alpha, beta, pw, pb), _ = _block_case(B, C, C, G)
    packed = ops.conv_block_pack(w, cb, alpha, beta, pw, pb)
    got = ops.conv_block(x, packed, C, variant=variant)
    assert torch.equal(got, ops.conv_block(x, packed, C, variant=variant | 1))


@pytest.mark.parametrize("B,M,K,G", [(3, 3, 24, 16), (2, 3, 417, 8), (1, 5, 7, 5), (1, 1, 3, 4)])
def test_pointwise_conv_matches_torch(B, M, K, G):
    from gfnet_amd import ops

    t = _rand(B, K, G, G, seed=11)
    w = _rand(M, K, seed=12, scale=K ** -0.5)
    b = _rand(M, seed=13)
    want = F.conv2d(t.double(), w.double().reshape(M, K, 1, 1), b.double())
    got = ops.pointwise_conv(t, w, b)
    err, mag = _maxerr(got, want)
    assert got.shape == want.shape and err <= 1e-5 * max(mag, 1.0), (err, mag)


def test_conv_block_rejects_bad_arguments():
    from gfnet_amd import ops
    from gfnet_amd._lib import GfnError

    x, (w, cb, alpha, beta, pw, pb), _ = _block_case(1, 8, 8, 8)
    packed = ops.conv_block_pack(w, cb, alpha, beta, pw, pb)
    with pytest.raises(GfnError):
        ops.conv_block(x, packed, 8, out=x)  # in place
    with pytest.raises(ValueError):
        ops.conv_block(x, packed, 40)  # packed for another shape
    with pytest.raises(GfnError):
        ops.pointwise_conv(x, _rand(17, 8), _rand(17))  # too many outputs for the small-M kernel
    with pytest.raises(GfnError):
        ops.conv_block(x, packed, 8, variant=4)


@pytest.mark.parametrize("feat,disp,r,G,B", [(8, 8, 0, 64, 3), (16, 16, 2, 32, 2), (32, 32, 4, 16, 2), (64, 64, 7, 8, 1),
                                             (64, 64, 6, 40, 1)])
def test_conv_stack_matches_torch_modules(feat, disp, r, G, B):
    """The whole stack of a real refiner configuration (random BatchNorm statistics) against the
    same nn modules run by torch in fp32 (MIOpen/ATen)."""
    from gfnet_amd.model.network import _refiner_for

    torch.manual_seed(5)
    ref = _refiner_for(feat, disp, r).cuda().eval()
    ref.amp = False
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
    C = ref.block1[0].in_channels
    d = _rand(B, C, G, G, seed=21)
    with torch.no_grad():
        want = ref.out_conv(ref.hidden_blocks(ref.block1(d.clone())))
        got = ref.conv_stack(d)
        got2 = ref.conv_stack(d, variant=1)
        ref.conv_precision = "fp16"
        got16 = ref.conv_stack(d)
        ref.conv_precision = "fp32"
        with torch.autocast("cuda", dtype=torch.float16):
            want16 = ref.out_conv(ref.hidden_blocks(ref.block1(d.clone())).float())
    err, mag = _maxerr(got, want)
    assert err <= 1e-4 * max(mag, 1.0), (err, mag)
    assert torch.equal(got, got2)
    with torch.no_grad():  # out_conv as its own kernel instead of multiplied into the last block
        ref.fold_out_conv = False
        unfolded = ref.conv_stack(d)
        ref.fold_out_conv = True
    err, _ = _maxerr(unfolded, want)
    assert err <= 1e-4 * max(mag, 1.0), (err, mag)
    # fp16 operands: closer to the fp32 result than torch's own fp16 autocast of the same modules is
    err16, _ = _maxerr(got16, want)
    err_amp, _ = _maxerr(want16, want)
    assert err16 <= 2e-2 * max(mag, 1.0), (err16, mag)
    assert err16 <= 2.0 * err_amp + 1e-6, (err16, err_amp)
    # amp: fp16 maps between the blocks too -- what torch's autocast does; as close to fp32 as torch's own autocast
    with torch.no_grad():
        ref.conv_precision = "amp"
        got_amp = ref.conv_stack(d)
        ref.conv_precision = "fp32"
    err_hm, _ = _maxerr(got_amp, want)
    assert got_amp.dtype == torch.float32 and got_amp.shape == want.shape
    assert err_hm <= 2e-2 * max(mag, 1.0), (err_hm, mag)
    assert err_hm <= 2.0 * err_amp + 1e-6, (err_hm, err_amp)
    # a changed running statistic invalidates the folded parameters
    with torch.no_grad():
        ref.block1[1].running_mean.add_(0.5)
        assert not torch.equal(ref.conv_stack(d), got)


@pytest.mark.parametrize("feat,disp,r,G,B", [(8, 8, 0, 24, 2), (16, 16, 2, 12, 1), (8, 6, 2, 10, 2)])
def test_conv_stack_matches_oracle(feat, disp, r, G, B):
    """HIP stack (C ABI) against the CPU oracle's restatement (pinned on the reference's golden G4 in
    tests/test_oracle_golden.py), same state_dict and input; G = 10 takes the two-pass kernels."""
    import oracle
    from gfnet_amd.model.network import _refiner_for

    torch.manual_seed(11)
    ref = _refiner_for(feat, disp, r).cuda().eval()
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.3)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
    C = ref.block1[0].in_channels
    d = _rand(B, C, G, G, seed=31)
    sd = {k: v.detach().cpu().numpy() for k, v in ref.state_dict().items()}
    want = oracle.conv_stack(d.cpu().numpy(), sd, variant="f64")
    with torch.no_grad():
        got = ref.conv_stack(d)
    err = float(np.abs(got.cpu().numpy().astype(np.float64) - want).max())
    assert err <= 1e-4 * max(float(np.abs(want).max()), 1.0), err


def test_training_mode_uses_torch_modules():
    from gfnet_amd.model.network import _refiner_for

    ref = _refiner_for(8, 8, 0).cuda()
    ref.train()
    assert not ref._hip_stack_supported()
    ref.eval()
    assert ref._hip_stack_supported()
    ref.conv_impl = "torch"
    assert not ref._hip_stack_supported()
