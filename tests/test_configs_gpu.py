"""GPU: the BASELINE.json configurations nobody had run in round 1 ("configs_untested"):
configs[2] googlemap 672x672 (pyramid sides 48/84/168/336, grids 48/48/96/192/384, num_itr = [2]*5 from map.json) and
configs[4] multi-scale 224 / 448 / 672 pyramids stored in fp16 -- the whole coarse-to-fine loop of both passes
(model/network.py:230-281, 326-349) checked per scale and per refiner iteration against the oracle run of the same loop,
plus full-batch (64-direction) local-correlation parity at the production shapes.  Tolerance 1e-4 (abs+rel)."""
import os
import sys

import numpy as np
import pytest
import torch

import oracle
from conftest import assert_close

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def host(t):
    torch.cuda.synchronize()
    return t.detach().float().cpu().numpy()


@pytest.mark.parametrize("size,num_itr,dtype,pairs", [(672, [2] * 5, torch.float32, 1), (224, [2] * 5, torch.float32, 1),
                                                      (224, [1] * 5, torch.float16, 1), (672, [1, 2, 1, 2, 1], torch.float16, 1),
                                                      (448, [1] * 5, torch.float32, 1), (448, [2] * 5, torch.float32, 1),
                                                      (224, [1, 2, 1, 1, 2], torch.float32, 4)])
def test_whole_path_per_scale_and_iteration_vs_oracle(size, num_itr, dtype, pairs):
    _whole_path_vs_oracle(size, num_itr, dtype, pairs, range(pairs))


def test_whole_path_at_the_metric_batch_vs_oracle():
    """the batch the metric is quoted on -- Scene(448, 32 pairs) = 64 directions, exactly what bench.py steps --
    through both passes and match_post, four pairs OF THAT BATCH (first, last and two seeded picks) against their own oracle walks:
    batch indexing at B = 32 / 64 directions of every kernel of the loop, not only of the local correlation."""
    rng = np.random.default_rng(5)
    which = sorted({0, 31, *(int(v) for v in rng.choice(np.arange(1, 31), 2, replace=False))})
    _whole_path_vs_oracle(448, [1] * 5, torch.float32, 32, which)


def _whole_path_vs_oracle(size, num_itr, dtype, pairs, which):
    """672-, 448- and 224-sized pyramids (not 448 pyramids in a bigger image) through forward_pyramids of both passes with the
    refiner iterations of map.json / basic.json; every flow / certainty the loop produces against the oracle's, then match_post.
    448 is BASELINE configs[1]'s own loop (model/network.py:230-281, 326-349 at 448 / 560); the 4-pair case (8 directions)
    checks the batch indexing of the whole loop: every pair against its own oracle walk.  Where a scale runs two iterations the
    eval-time zeroing decision itself (network.py:264-265) is compared too: outside the band around the threshold no element
    may be zeroed on the device that the oracle moved."""
    from gfnet_amd import _synthetic as synthetic
    from oracle.scene import cpu_pair

    dev = torch.device("cuda", 0)
    sc = synthetic.Scene(size, pairs, num_itr, dtype, "off", dev, rank=0)
    m = sc.model
    with torch.inference_mode():
        m.train(False)
        cor = m.forward_pyramids(sc.pyr[0], sc.pyr[1], (size, size), symmetric=True)
        m.num_grid_up, m.radius_up, m.num_itr_up = m.upsample_grids(sc.up)
        sf = float(np.sqrt(sc.up * sc.up / (size * size)))
        cup = m.forward_pyramids(sc.pyr_up[0], sc.pyr_up[1], (sc.up, sc.up), symmetric=True, upsample=True, scale_factor=sf,
                                 pre_corresps=cor["1"][m.num_itr[-1]])
        warp, cert = m._finish_match(cor, cup, batched=True)
    to_np = lambda p: {s: t.float().cpu().numpy() for s, t in p.items()}  # noqa: E731
    np_gt = {G: t.cpu().numpy() for G, t in sc.gt.items()}
    np_noise = {G: [n.numpy() for n in ns] for G, ns in sc.noise.items()}
    npyr, nup = (to_np(sc.pyr[0]), to_np(sc.pyr[1])), (to_np(sc.pyr_up[0]), to_np(sc.pyr_up[1]))

    # The eval-time rule of network.py:264-265 zeroes a displacement that repeats the previous one to 1e-6: discontinuous, and
    # with the stand-in refiner (displacements = differences of nearby floats, a few thousand representable values) a handful
    # of cells per map repeat EXACTLY on one side and miss by one ulp on the other.  Those cells (rel within 10x of the
    # threshold in the oracle) are compared with the tolerance of one displacement instead; they must stay rare.
    def check(got, ref, rel, what):
        ref = np.asarray(ref)
        amb = (rel < 1e-5).any(axis=1, keepdims=True) & np.ones_like(ref, bool)
        assert amb.mean() < 1e-3, f"{what}: {amb.mean():.2e} of the cells sit on the zeroing threshold"
        assert_close(np.where(amb, ref, got), ref, 1e-4, what)
        assert np.abs(got - ref)[amb].max(initial=0) < 8.0 / size, what  # a flipped cell is off by one displacement (~ the flow noise)
        return amb

    for b in which:
        rows = [b, b + pairs]  # the pair's two directions in the symmetric batch
        r1, r2, warp_o, cert_o = cpu_pair(sc, b, npyr, nup, np_gt, np_noise, seed=0, return_all=True)
        last_amb = None
        for res, corr, sz, scales in ((r1, cor, size, synthetic.SCALES), (r2, cup, sc.up, synthetic.SCALES[1:])):
            for s in scales:
                prev = None
                for itr in corr[s]:
                    got = host(corr[s][itr]["flow"])[rows]
                    ref, rel = np.asarray(res[(s, itr)][0]), np.asarray(res[(s, itr)][2])
                    last_amb = check(got, ref, rel, f"{sz} pair {b}: flow {s}.{itr}")
                    assert_close(host(corr[s][itr]["certainty"])[rows], res[(s, itr)][1], 1e-4, f"{sz} pair {b}: cert {s}.{itr}")
                    if prev is not None:
                        # iteration >= 2: the zeroing decision itself.  An element is zeroed iff its displacement repeats the
                        # previous one to 1e-6; seen from outside, a zeroed element's flow does not move.  Outside the band
                        # around the threshold (oracle ratio > 1e-5: nothing is zeroed there -- inside it the two sides may
                        # differ, see above, and an exact repeat, ratio 0, is the band's other edge) every element whose
                        # oracle flow moved by more than a few ulps must have moved on the device as well, element for element.
                        clear = (rel > 1e-5) & (np.abs(ref - prev[1]) > 1e-5)  # (a NaN ratio, 0 / 0, compares False)
                        assert clear.mean() > 0.9, f"{sz} pair {b}: {s}.{itr}: the mask check would be vacuous"
                        assert not np.any((got == prev[0])[clear]), f"{sz} pair {b}: zeroing mask {s}.{itr}"
                    prev = (got, ref)
        # match_post: the cells that flipped in the very last update are excluded from the warp (both halves of the symmetric layout)
        amb_w = np.concatenate((last_amb[:1, 0], last_amb[1:, 0]), axis=2)[..., None] & np.ones_like(warp_o, bool)
        assert_close(np.where(amb_w, warp_o, host(warp)[b:b + 1]), warp_o, 1e-4, f"pair {b}: warp")
        assert_close(np.where(amb_w[..., 0], cert_o, host(cert)[b:b + 1]), cert_o, 1e-4, f"pair {b}: certainty")


def _bench_flows(B, G, S, seed):
    """Flows like the bench's: true warps of 15 % corner-perturbation homographies (both directions) + 0.25-px noise."""
    from gfnet_amd import _synthetic as synthetic

    gen = torch.Generator().manual_seed(seed)
    H = synthetic.random_homographies(B // 2, S, gen)
    f = torch.cat((synthetic.warp_grid(H, G, S, "cpu"), synthetic.warp_grid(np.linalg.inv(H), G, S, "cpu"))).permute(0, 3, 1, 2)
    f = f + torch.randn(B, 2, G, G, generator=gen) * (0.5 / S)
    return f.contiguous().numpy().astype(np.float32)


@pytest.mark.parametrize("c,hs,G,r,S", [(32, 112, 64, 4, 448), (16, 224, 128, 2, 448), (32, 168, 96, 4, 672)])
def test_full_batch_local_correlation_vs_oracle(c, hs, G, r, S):
    """configs[1] at its real batch (64 directions: grid size, XCD remap, plan / halves / second-launch lists all differ
    from the B = 2 cases) and configs[2]'s scale-4 shape at 32 directions, whole tensor against the oracle."""
    import synth
    from gfnet_amd.utils.local_correlation import local_correlation

    B = 64 if S == 448 else 32
    f0 = synth.lattice_normalish((B, c, G, G), 701 + r)
    f1 = synth.lattice_normalish((B, c, hs, hs), 702 + r)
    flow = _bench_flows(B, G, S, 703 + r)
    out = local_correlation((B, c, hs, hs), torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda(), r, G, flow=torch.from_numpy(flow).cuda())
    ref = oracle.local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow)
    assert_close(host(out), ref, 1e-4, f"full batch c{c} hs{hs} G{G} r{r}")
    args = ((B, c, hs, hs), torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda(), r, G)
    old = local_correlation(*args, flow=torch.from_numpy(flow).cuda(), _variant=2)
    lean = local_correlation(*args, flow=torch.from_numpy(flow).cuda(), _variant=4)  # round-2 lean kernel: fp32 FMA D-stage
    np.testing.assert_array_equal(host(lean), host(old))
    assert_close(host(out), host(old), 1e-4, "default path (matrix-core D-stage where enabled) vs the fp32 FMA kernels")


def _raw_softargmax_flows(B, S, seed):
    """The flows scale 16 really sees: the raw output of corr_softargmax on a synthetic scene's coarsest maps (near one-hot
    soft-argmax: cell centres of the other image, which scatter a third of the r = 7 tiles past the matrix-core kernel's
    accumulators).  Returns (flow (B,2,G,G) numpy, f-maps side)."""
    from gfnet_amd import _synthetic as synthetic, ops

    gen_cpu = torch.Generator().manual_seed(seed)
    gen = torch.Generator(device="cuda").manual_seed(seed + 1)
    H = synthetic.random_homographies(B // 2, S, gen_cpu)
    pa, pb = synthetic.make_pyramids(H, S, ["16"], torch.device("cuda"), gen, torch.float32)
    flow = ops.corr_softargmax(pa["16"], pb["16"], symmetric=True)
    return host(flow).astype(np.float32)


@pytest.mark.parametrize("c,hs,G,r,S", [(64, 32, 32, 7, 448), (64, 56, 32, 6, 448), (64, 70, 40, 6, 448), (64, 48, 48, 7, 672)])
def test_full_batch_large_windows_vs_oracle(c, hs, G, r, S):
    """The r >= 5 matrix-core kernel (csrc/local_corr_mq.h, the default path of basic.json's radius [7, 6, ...]) at the production
    batch: 64 directions at 448 (32 at 672), whole tensor against the oracle.  The r = 7 cases run on the raw soft-argmax flows of
    a synthetic scene -- the flows that send a third of the tiles through the fp32 routine inside the launch and some to the
    second launch -- and all three routes of a tile must have been taken there (counters in the scratch header)."""
    import synth
    from gfnet_amd import _lib
    from gfnet_amd.utils.local_correlation import local_correlation

    B = 64 if S == 448 else 32
    f0 = synth.lattice_normalish((B, c, G, G), 801 + r)
    f1 = synth.lattice_normalish((B, c, hs, hs), 802 + r)
    flow = _raw_softargmax_flows(B, S, 803 + r) if r == 7 else _bench_flows(B, G, S, 803 + r)
    assert flow.shape == (B, 2, G, G)
    out = local_correlation((B, c, hs, hs), torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda(), r, G, flow=torch.from_numpy(flow).cuda())
    dev = torch.device("cuda", torch.cuda.current_device())
    hdr = _lib.scratch(dev, int(_lib.lib().gfn_local_corr_scratch_bytes(B, G)))[:8].cpu().numpy()  # csrc/local_corr.hip kTodoHdr
    ref = oracle.local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow)
    assert_close(host(out), ref, 1e-4, f"full batch c{c} hs{hs} G{G} r{r}")
    tiles = B * ((G + 1) // 2) * ((G + 15) // 16)
    listed, in_launch = int(hdr[3]), int(hdr[7])
    if r == 7:
        assert listed > 0, "no tile went to the second launch"
        assert in_launch > 0, "no tile took the fp32 routine inside the launch"
        assert listed + in_launch < tiles, "no tile took the matrix-core path"
    old = local_correlation((B, c, hs, hs), torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda(), r, G, flow=torch.from_numpy(flow).cuda(),
                            _variant=2)
    assert_close(host(out), host(old), 1e-4, "matrix-core path vs the fp32 FMA kernel")


@pytest.mark.parametrize("G,S", [(48, 672), (96, 672), (192, 672), (384, 672), (16, 224), (32, 224), (64, 224), (128, 224)])
def test_grid_ops_on_the_672_and_224_grids(G, S):
    """refiner_input, flow_update, the inter-scale resize and match_post on the grids of configs[2] / configs[4] (round 1
    only ran them on the 448 grids)."""
    import synth
    from gfnet_amd import ops

    B, c = 2, 8
    hs = max(S // (448 // 56) // 2, 8)
    x = synth.lattice_normalish((B, c, hs, hs), 801)
    y = synth.lattice_normalish((B, c, hs, hs), 802)
    flow = _bench_flows(2 * B, G, S, 803)
    w = synth.lattice_uniform((6, 2), 804)
    bias = synth.lattice_uniform((6,), 805)
    d = ops.refiner_input(G, torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(flow).cuda(), torch.from_numpy(w).cuda(),
                          torch.from_numpy(bias).cuda(), 0, scale_factor=1.5, corr_in_other=False)
    xs, ys = np.concatenate((x, y)), np.concatenate((y, x))
    ref = oracle.refiner_input(G, xs, ys, flow, w, bias, 0, scale_factor=1.5, corr_in_other=False)
    assert_close(host(d), ref, 1e-4, f"refiner_input G{G}")
    cert = synth.lattice_uniform((2 * B, 1, G, G), 806)
    dl = synth.lattice_normalish((2 * B, 3, G, G), 807)
    prev = np.full((2 * B, 2, G, G), 1e-7, np.float32)
    tprev = torch.from_numpy(prev.copy()).cuda()
    fo, co = ops.flow_update(torch.from_numpy(flow).cuda(), torch.from_numpy(cert).cuda(), torch.from_numpy(dl[:, :2].copy()).cuda(),
                             torch.from_numpy(dl[:, 2:3].copy()).cuda(), tprev, 4, S, S)
    rf, rc, rd = oracle.flow_update(flow, cert, dl[:, :2], dl[:, 2:3], prev, 4, S, S)
    assert_close(host(fo), rf, 1e-5, "flow_update flow")
    assert_close(host(co), rc, 1e-5, "flow_update cert")
    assert_close(host(tprev), rd, 1e-5, "flow_update displacement")
    G2 = 2 * G
    a, b2 = ops.interpolate_bilinear_pair(fo, co, G2)
    assert_close(host(a), oracle.interpolate_bilinear(rf, G2), 1e-5, "resize flow")
    assert_close(host(b2), oracle.interpolate_bilinear(rc, G2), 1e-5, "resize cert")
    c16 = synth.lattice_normalish((2 * B, 1, max(G // 8, 2), max(G // 8, 2)), 808)
    warp, cc = ops.match_post(fo, co, torch.from_numpy(c16).cuda(), symmetric=True)
    wo, co_ = oracle.match_post(rf, rc, c16, symmetric=True, attenuate_cert=True)
    assert_close(host(warp), wo, 1e-5, "match_post warp")
    assert_close(host(cc), co_, 1e-5, "match_post certainty")


# ---- configs[4]: feature pyramids stored in fp16 are read as stored (no widened copy) ------------------------------------
@pytest.mark.parametrize("c,hs,G,r,kind", [(32, 112, 64, 4, "bench"), (16, 224, 128, 2, "bench"), (64, 56, 32, 6, "bench"), (64, 32, 32, 7, "random"),
                                           (32, 70, 40, 4, "bench"), (16, 60, 40, 3, "random"), (8, 20, 12, 2, "bench"), (32, 168, 96, 4, "zoom")])
@pytest.mark.parametrize("variant", [0, 2])
def test_fp16_feature_maps_are_read_natively_and_match_the_widened_copy(c, hs, G, r, kind, variant):
    """All arithmetic stays fp32, so reading fp16 maps directly must give bit for bit what the fp32 kernels give on the
    widened copy (lean path, round-1 tile kernel, second launch, general kernel), and that is the oracle on rounded inputs."""
    import synth
    from gfnet_amd.utils.local_correlation import local_correlation

    B = 4
    f0 = torch.from_numpy(synth.lattice_normalish((B, c, G, G), 901)).cuda()
    f1h = torch.from_numpy(synth.lattice_normalish((B, c, hs, hs), 902) * np.float32(1.37)).cuda().half()  # off the fp16 lattice before rounding
    if kind == "bench":
        flow = _bench_flows(B, G, 448, 903)
    elif kind == "zoom":
        flow = synth.homography_flow(B, G, 904, scale=1.6)
    else:
        flow = 1.2 * synth.lattice_uniform((B, 2, G, G), 905)
    fl = torch.from_numpy(flow).cuda()
    got = local_correlation((B, c, hs, hs), f0, f1h, r, G, flow=fl, _variant=variant)
    ref = local_correlation((B, c, hs, hs), f0, f1h.float(), r, G, flow=fl, _variant=variant)
    assert got.dtype == torch.float32
    np.testing.assert_array_equal(host(got), host(ref))
    if variant == 0:
        want = oracle.local_correlation((B, c, hs, hs), host(f0), host(f1h), r, G, flow=flow)
        assert_close(host(got), want, 1e-4, "fp16 f1 vs oracle on the rounded inputs")


def test_fp16_refiner_input_and_corr_softargmax_match_the_widened_copy():
    import synth
    from gfnet_amd import ops

    B, c, hs, G = 2, 16, 57, 24  # odd row length: 2-byte aligned pair gathers
    x = torch.from_numpy(synth.lattice_normalish((B, c, hs, hs), 911) * np.float32(0.77)).cuda().half()
    y = torch.from_numpy(synth.lattice_normalish((B, c, hs, hs), 912) * np.float32(0.77)).cuda().half()
    flow = torch.from_numpy(_bench_flows(2 * B, G, 448, 913)).cuda()
    w = torch.from_numpy(synth.lattice_uniform((6, 2), 914)).cuda()
    bias = torch.from_numpy(synth.lattice_uniform((6,), 915)).cuda()
    for r in (0, 2):
        d16 = ops.refiner_input(G, x, y, flow, w, bias, r, scale_factor=1.25, corr_in_other=r > 0)
        d32 = ops.refiner_input(G, x.float(), y.float(), flow, w, bias, r, scale_factor=1.25, corr_in_other=r > 0)
        np.testing.assert_array_equal(host(d16), host(d32))
    xs, ys = np.concatenate((host(x), host(y))), np.concatenate((host(y), host(x)))
    assert_close(host(d16), oracle.refiner_input(G, xs, ys, host(flow), host(w), host(bias), 2, scale_factor=1.25), 1e-4, "refiner_input fp16")
    f0 = torch.from_numpy(synth.lattice_normalish((3, 64, 32, 32), 921) * np.float32(0.31)).cuda().half()
    f1 = torch.from_numpy(synth.lattice_normalish((3, 64, 32, 32), 922) * np.float32(0.31)).cuda().half()
    for sym in (False, True):
        a, b = ops.corr_softargmax(f0, f1, symmetric=sym), ops.corr_softargmax(f0.float(), f1.float(), symmetric=sym)
        np.testing.assert_array_equal(host(a), host(b))
    assert_close(host(ops.corr_softargmax(f0, f1)), oracle.corr_softargmax(host(f0), host(f1)), 1e-4, "corr_softargmax fp16")


def test_a_scene_step_replayed_from_a_hipgraph_equals_the_eager_step():
    """Round 4: a whole step (both passes of the coarse-to-fine loop, sampling, solve: ~300 launches) captured once into a hipGraph
    (Scene.capture: torch.cuda.graph around the C-ABI launches, which allocate nothing and run on the caller's stream) and replayed:
    H and the sampled matches are bit-identical to the eager step with the same seeds, replay after replay.
    (The comparisons run on the capture stream: on ROCm 7.2 default-stream work that reads a replay's outputs together with tensors of the
    eager steps, between two replays of the full-step graph, ends a later replay with a memory fault -- reproduced with
    tools/dbg_graph.py, cause not found)"""
    from gfnet_amd._synthetic import Scene

    dev = torch.device("cuda", torch.cuda.current_device())
    with torch.inference_mode(False):
        sc = Scene(224, 2, [1] * 5, torch.float16, "off", dev, 0)
    with torch.inference_mode():
        torch.manual_seed(7)  # the sampler's seeds come from torch's CPU generator: three steps eagerly, the third is the reference
        for _ in range(3):
            He, ge = sc.step(5)
        He, ge = He.clone(), ge.clone()
        torch.cuda.synchronize()
        torch.manual_seed(7)  # capture() runs two warm-up steps, then captures the third
        Hg, gg = sc.capture(5, warmup=2)
        for _ in range(3):
            sc.replay()
            with torch.cuda.stream(sc._gstream):
                same = bool(torch.equal(Hg, He)) and bool(torch.equal(gg, ge))
                finite = bool(torch.isfinite(Hg).all())
            assert same and finite


def test_the_two_stage_hipgraph_replay_equals_the_eager_steps():
    """Round 4: match and finish captured apart, twice each (Scene.capture_pipelined: a step's sampling + solve replays under the next
    step's matching; the two copies of a stage alternate so that a match never overwrites what a finish still reads).  Copy 0 holds the
    sampler draws of the third eager step with the same generator seed, copy 1 those of the fourth; six replays in a row return, in
    turn, exactly those two eager results.  (Comparisons on the finish stream, as in the one-graph test above.)"""
    from gfnet_amd._synthetic import Scene

    dev = torch.device("cuda", torch.cuda.current_device())
    with torch.inference_mode(False):
        sc = Scene(224, 2, [1] * 5, torch.float16, "off", dev, 0)
    with torch.inference_mode():
        torch.manual_seed(11)
        eager = []
        for _ in range(4):
            He, ge = sc.step(5)
            eager.append((He.clone(), ge.clone()))
        torch.cuda.synchronize()
        torch.manual_seed(11)  # capture_pipelined: two warm-up finishes, then the two captured copies
        sc.capture_pipelined(5, warmup=2)
        for rep in range(6):
            (Hg, gg), done = sc.replay_pipelined()
            He, ge = eager[2 + (rep & 1)]
            with torch.cuda.stream(sc._fs):
                same = bool(torch.equal(Hg, He)) and bool(torch.equal(gg, ge))
                finite = bool(torch.isfinite(Hg).all())
            assert same and finite, f"replay {rep}"
        torch.cuda.synchronize()


def test_three_stage_streaming_of_batches_equals_the_one_stream_steps():
    """Round 4: bench.py's default for one-scene workloads runs a step as three stages on three HIP streams (first pass | refinement
    pass + post-processing | sampling + solve: GFNet.match_first_pass / match_second_pass), a step's later stages beside the next
    steps' earlier ones.  Every step's H and sampled matches are bit-identical to the same steps on one stream (same seeds, same
    generator state), step after step -- the stages only meet through events."""
    import bench
    from gfnet_amd._synthetic import Scene

    dev = torch.device("cuda", torch.cuda.current_device())
    with torch.inference_mode(False):
        sc = Scene(224, 3, [1] * 5, torch.float32, "off", dev, 0)
    with torch.inference_mode():
        torch.manual_seed(3)
        ref = []
        for i in range(4):
            H, good = sc.step(i)
            ref.append((H.clone(), good.clone()))
        torch.cuda.synchronize()
        runner = bench.SceneRunner([sc], pipeline=True, stages=3)
        assert runner.stages3 and len({s.cuda_stream for s in runner.streams[0]}) == 3
        torch.manual_seed(3)
        outs = [runner.step(i)[0] for i in range(4)]  # four steps in flight back to back
        torch.cuda.synchronize()
        for i, ((H, good), (Hr, gr)) in enumerate(zip(outs, ref)):
            assert torch.equal(H, Hr) and torch.equal(good, gr), f"step {i}"


def test_three_stage_hipgraph_replay_and_graphs_of_two_scenes_on_one_stream():
    """Round 4: (a) Scene.capture_pipelined(stages=3) -- first pass | refinement pass + post-processing | sampling + solve as graphs
    on three streams, two copies each -- replays the eager steps bit for bit; (b) two scenes of different sizes captured one after
    the other on ONE stream: the second capture outgrows the stream's scratch buffer, which used to free the buffer the first graph's
    kernels still name (memory fault on replay); outgrown buffers are retired now (gfnet_amd/_lib.py) and both graphs replay."""
    from gfnet_amd import parallel
    from gfnet_amd._synthetic import Scene

    dev = torch.device("cuda", torch.cuda.current_device())
    with torch.inference_mode(False):
        sc = Scene(224, 2, [1] * 5, torch.float16, "off", dev, 0)
        small = Scene(224, 1, [1] * 5, torch.float32, "off", dev, 0)
        big = Scene(448, 2, [1] * 5, torch.float32, "off", dev, 0)
    with torch.inference_mode():
        torch.manual_seed(11)
        eager = []
        for _ in range(4):
            He, ge = sc.step(5)
            eager.append((He.clone(), ge.clone()))
        torch.cuda.synchronize()
        torch.manual_seed(11)
        sc.capture_pipelined(5, warmup=2, streams=parallel.concurrent_streams(3), stages=3)
        for rep in range(6):
            (Hg, gg), done = sc.replay_pipelined()
            He, ge = eager[2 + (rep & 1)]
            with torch.cuda.stream(sc._fs):
                same = bool(torch.equal(Hg, He)) and bool(torch.equal(gg, ge))
            assert same, f"replay {rep}"
        torch.cuda.synchronize()
        # (b)
        st = torch.cuda.Stream()
        torch.manual_seed(2)
        Hs, _ = small.capture(1, stream=st)
        small.replay()  # (a capture runs nothing: the static outputs are filled by the first replay)
        with torch.cuda.stream(st):
            first = Hs.clone()
        torch.manual_seed(2)
        Hb, _ = big.capture(1, stream=st)   # needs a larger scratch buffer on the same stream
        for _ in range(3):
            small.replay()
            big.replay()
            with torch.cuda.stream(st):
                ok = bool(torch.equal(Hs, first)) and bool(torch.isfinite(Hb).all())
            assert ok
        torch.cuda.synchronize()
