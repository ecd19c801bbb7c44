"""Synthetic workloads of the post-backbone hot path: feature pyramids seen through random homographies, the
true warps on every grid, and a stand-in for the refiners' conv stacks.  Shared by bench.py, tools/ and the whole-path parity
tests (tests/test_configs_gpu.py), so that an edit to the bench cannot silently change what the tests check.
Nothing here touches oracle/: the CPU walk of a scene lives in oracle/scene.py.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

FEAT = {"16": 64, "8": 64, "4": 32, "2": 16, "1": 8}
DISP = {"16": 64, "8": 64, "4": 32, "2": 16, "1": 8}
RADIUS = [7, 6, 4, 2, 0]
SCALES = ["16", "8", "4", "2", "1"]
FLOW_NOISE_PX = 0.5    # true flow + N(0, (0.5/S)^2) in normalised units
# Stress flows (bench.py --flows): what the refiner of a scale hands to the next one when it is NOT a near-perfect
# matcher.  "noisy": the per-scale noise of the stand-in increment is several image pixels at the coarse scales (a scale's local
# correlation sees the previous scale's output, bilinearly resized); "random": the scale-8 refiner returns flows uniform in
# [-0.9, 0.9] (every window of the scale-4 call scattered: all of its tiles go through the second launch's gather path), the
# later scales pull the flow back onto the truth.
FLOW_MODES = {"true": {"16": 1, "8": 1, "4": 1, "2": 1, "1": 1},
              "noisy": {"16": 8, "8": 8, "4": 4, "2": 2, "1": 1},      # x 0.5 px: 4 / 4 / 2 / 1 / 0.5 image pixels
              "random": {"16": 1, "8": None, "4": 1, "2": 1, "1": 1}}

WORKLOADS = {
    "448b32": {"sizes": [448], "pairs": 32, "num_itr": [1] * 5, "dtype": "fp32", "cpu_pairs": 2,
               "label": "448x448 batch=32 synthetic pairs per GPU, local_correlation radius 7/6/4/2 (BASELINE configs[1])"},
    "672b16": {"sizes": [672], "pairs": 16, "num_itr": [2] * 5, "dtype": "fp32", "cpu_pairs": 1,
               "label": "googlemap 672x672 batch=16 per GPU, num_itr=[2]*5 (gfnet_configs/map.json), grids 48/48/96/192/384 "
                        "(BASELINE configs[2])"},
    "pyr-fp16": {"sizes": [224, 448, 672], "pairs": 8, "num_itr": [1] * 5, "dtype": "fp16", "cpu_pairs": 1,
                 "label": "multi-scale 224/448/672 pyramids stored in fp16, 8 pairs per size and step, streamed KDE + device solve "
                          "(BASELINE configs[4])"},
}


def grids_for(size):
    """num_grid of a pass at image size `size`: network.py:329's rule, [hs/14, 2x, 4x, 8x] with the coarsest repeated for
    scale 16 (basic.json's [32,32,64,128,256] at 448)."""
    g = int(size / 14)
    return [g, g, 2 * g, 4 * g, 8 * g]


def side_of(scale, size):
    return size // 14 if scale == "16" else size // int(scale)  # network.py:185-198: 32/56/112/224/448 at 448


def random_homographies(n, size, gen):
    """4-corner perturbation U(-0.15 S, 0.15 S), as (n,3,3) float64."""
    src = np.array([[0, 0], [size - 1, 0], [size - 1, size - 1], [0, size - 1]], np.float64)
    out = []
    for _ in range(n):
        dst = src + (torch.rand(4, 2, generator=gen, dtype=torch.float64).numpy() * 2 - 1) * 0.15 * size
        A = []
        for (x, y), (u, v) in zip(src, dst):
            A.append([x, y, 1, 0, 0, 0, -u * x, -u * y, -u])
            A.append([0, 0, 0, x, y, 1, -v * x, -v * y, -v])
        h = np.linalg.svd(np.array(A))[2][-1].reshape(3, 3)
        out.append(h / h[2, 2])
    return np.stack(out)


def warp_grid(H, side, size, device):
    """Normalised sampling grid (n,side,side,2): where each cell centre of a side x side map of the
    A image lands in the B image under H (pixel coordinates of a size x size image)."""
    lin = (torch.arange(side, dtype=torch.float64) * 2 + 1) / side - 1
    gy, gx = torch.meshgrid(lin, lin, indexing="ij")
    px, py = (size - 1) * (gx + 1) / 2, (size - 1) * (gy + 1) / 2
    Ht = torch.from_numpy(H)
    w = Ht[:, 2, 0, None, None] * px + Ht[:, 2, 1, None, None] * py + Ht[:, 2, 2, None, None]
    u = (Ht[:, 0, 0, None, None] * px + Ht[:, 0, 1, None, None] * py + Ht[:, 0, 2, None, None]) / w
    v = (Ht[:, 1, 0, None, None] * px + Ht[:, 1, 1, None, None] * py + Ht[:, 1, 2, None, None]) / w
    return torch.stack((2 * u / (size - 1) - 1, 2 * v / (size - 1) - 1), -1).float().to(device)


def make_pyramids(H, size, scales, device, gen, dtype):
    """Synthetic feature pyramids: B-image features = smoothed noise (amplitude 2), A-image features =
    the B features seen through H + 0.1 noise, so correlation peaks and flows are meaningful."""
    n = H.shape[0]
    pa, pb = {}, {}
    for s in scales:
        side, c = side_of(s, size), FEAT[s]
        fb = F.avg_pool2d(torch.randn(n, c, side, side, device=device, generator=gen), 3, 1, 1) * 6.0
        fa = F.grid_sample(fb, warp_grid(H, side, size, device), mode="bilinear", padding_mode="zeros", align_corners=False)
        fa = fa + 0.1 * torch.randn(n, c, side, side, device=device, generator=gen)
        pa[s], pb[s] = fa.to(dtype).contiguous(), fb.to(dtype).contiguous()
    return pa, pb


class StandInRefiner(nn.Module):
    """The HIP part of ConvRefiner.forward (network.py:533-558) followed by a stand-in for the conv
    stack (network.py:560-563): like a trained refiner it returns the increment that moves the flow onto the
    true warp plus sub-pixel noise (N(0, (0.5/S)^2), a fresh seeded realisation per refiner iteration) and a
    constant certainty increment.  One torch elementwise op; everything else is the real path.  With
    --conv-stack the reference's conv stack runs too (random-init, its output weighted 0)."""

    def __init__(self, feat, disp, radius, scale, targets, num_itr, conv_stack="off"):
        super().__init__()
        from gfnet_amd.model.network import ConvRefiner, _refiner_for

        K = (2 * radius + 1) ** 2 if radius > 0 else 0
        dim = 2 * feat + disp + K
        if conv_stack == "off":
            self.inner = ConvRefiner(dim, dim, 3, kernel_size=5, dw=True, hidden_blocks=0, displacement_emb="linear",
                                     displacement_emb_dim=disp, local_corr_num=radius, corr_in_other=radius > 0)
        else:
            self.inner = _refiner_for(feat, disp, radius)
            self.inner.conv_precision = conv_stack
        self.conv_stack = conv_stack
        self.scale, self.num_itr = scale, num_itr
        self.targets = targets  # {num_grid: ([k * (true flow + noise_itr) for itr], k)}
        self._cert, self._calls = {}, {}

    supports_reuse_d = True  # GFNet.forward_pyramids: later iterations at a scale keep the grid_feature planes

    def forward(self, num_grid, x, y, flow, scale_factor=1, reuse_d=None):
        prev = reuse_d[0] if reuse_d is not None else None
        reusable = reuse_d is not None and self.inner.may_reuse_d(x, y, flow)
        d, lc = self.inner.assemble(num_grid, x, y, flow, scale_factor, reuse=prev if reusable else None)
        if reuse_d is not None:
            reuse_d[0] = d if reusable else None
        tk, k = self.targets[num_grid]
        itr = self._calls.get(num_grid, 0)
        self._calls[num_grid] = (itr + 1) % self.num_itr
        delta = torch.add(tk[itr], flow, alpha=-k)  # (gt + noise - flow) * k in one launch; k undone by network.py:262-263
        if num_grid not in self._cert:
            self._cert[num_grid] = torch.full((flow.shape[0], 1, num_grid, num_grid), 1.0, device=flow.device)
        cert = self._cert[num_grid]
        if self.conv_stack != "off":
            out = self.inner.conv_stack(d)
            delta = torch.addcmul(delta, out[:, :2], torch.zeros((), device=d.device))
            cert = torch.addcmul(cert, out[:, 2:3], torch.zeros((), device=d.device))
        return delta, cert, lc


class Scene:
    """Everything one image size needs: pyramids of both passes, true warps + noise on every grid, the model."""

    def __init__(self, size, pairs, num_itr, dtype, conv_stack, dev, rank, upsample=True, flows="true"):
        from gfnet_amd.model.network import GFNet

        self.flows = flows
        mult = FLOW_MODES[flows]

        self.size, self.up, self.B, self.num_itr = size, int(size * 1.25), pairs, num_itr
        gen_cpu = torch.Generator().manual_seed(1000 + rank + 7 * size)
        gen = torch.Generator(device=dev).manual_seed(2000 + rank + 7 * size)
        S0, S1 = self.size, self.up
        self.H = random_homographies(pairs, S0, gen_cpu)
        self.pyr = make_pyramids(self.H, S0, SCALES, dev, gen, dtype)
        Hup = np.stack([np.diag([S1 / S0, S1 / S0, 1.0]) @ h @ np.diag([S0 / S1, S0 / S1, 1.0]) for h in self.H])
        self.pyr_up = make_pyramids(Hup, S1, SCALES[1:], dev, gen, dtype) if upsample else (None, None)
        self.grids, self.grids_up = grids_for(S0), grids_for(S1)[1:]
        # true normalised warps on every grid the two passes use (A->B for the first B rows, B->A after), plus the
        # per-iteration noise realisations (generated on the CPU so that the oracle leg sees the same bits)
        self.gt, self.noise = {}, {}
        Hinv, Hupinv = np.linalg.inv(self.H), np.linalg.inv(Hup)
        passes = [(self.grids, self.H, Hinv, S0)] + ([(self.grids_up, Hup, Hupinv, S1)] if upsample else [])
        for grids, Hf, Hb, S in passes:
            for G in set(grids):
                self.gt[G] = torch.cat((warp_grid(Hf, G, S, dev), warp_grid(Hb, G, S, dev))).permute(0, 3, 1, 2).contiguous()
                self.noise[G] = [torch.randn(2 * pairs, 2, G, G, generator=gen_cpu) * (FLOW_NOISE_PX / S) for _ in range(max(num_itr))]
        targets = {s: {} for s in SCALES}  # per scale: {num_grid: ([k * (true flow + noise) per iteration], k)}
        for i, s in enumerate(SCALES):
            uses = [(self.grids[i], S0)] + ([(self.grids_up[i - 1], S1)] if upsample and i >= 1 else [])
            for G, S in uses:
                k = 4.0 * S / int(s)  # undone by network.py:262-263's scale / (4 * W0)
                if mult[s] is None:   # uniform flows (seeded): the next scale's windows land anywhere in the map
                    targets[s][G] = ([(torch.rand(self.gt[G].shape, generator=gen_cpu).to(dev) * 1.8 - 0.9) * k for _ in range(num_itr[i])], k)
                else:
                    targets[s][G] = ([(self.gt[G] + n.to(dev) * float(mult[s])) * k for n in self.noise[G][:num_itr[i]]], k)
        refiners = nn.ModuleDict({s: StandInRefiner(FEAT[s], DISP[s], RADIUS[i], int(s), targets[s], num_itr[i], conv_stack)
                                  for i, s in enumerate(SCALES)})
        conf = {"encoder_cfg": {"feat_chs": [64, 32, 16, 8]},
                "matcher": {"num_grid": self.grids, "radius": RADIUS, "displacement_dim": [64, 64, 32, 16, 8], "num_itr": num_itr}}
        self.model = GFNet(conf, initial_res=(S0, S0), upsample_res=(S1, S1), symmetric=True, upsample_preds=upsample,
                           attenuate_cert=True, conv_refiner=refiners).to(dev).eval()
        self.sizes = (S0, S0, S0, S0)
        self.roofline_key = f"local_corr_c32_h{side_of('4', S0)}_g{self.grids[2]}_r4"

    def match(self):
        """Both passes of the coarse-to-fine loop + match post-processing: (warp, certainty) of the batch."""
        return self.model.match_pyramids(self.pyr[0], self.pyr[1], self.pyr_up[0], self.pyr_up[1], batched=True)

    def match_first(self):
        """First pass only (its corresps): GFNet.match_first_pass."""
        return self.model.match_first_pass(self.pyr[0], self.pyr[1])

    def match_second(self, corresps):
        """Refinement pass + post-processing on a first pass's corresps: (warp, certainty)."""
        return self.model.match_second_pass(corresps, self.pyr_up[0], self.pyr_up[1], batched=True)

    def finish(self, warp, cert, seed):
        """Balanced sampling + homography solve on a batch's (warp, certainty)."""
        from gfnet_amd.estimation import estimate_homographies
        from gfnet_amd.model.network import sample_batched

        good, _ = sample_batched(self.model, warp, cert, 5000)
        Hl = estimate_homographies(good, self.sizes, iters=self.model.ransac_iters, seed=seed)
        return Hl, good

    def step(self, seed):
        warp, cert = self.match()
        return self.finish(warp, cert, seed)

    # ---- a step as a hipGraph -----------------------------------------------------------------------------------------------------
    # The small scenes of the pyramid workload are bound by the host's launch rate (~300 launches of 15-60 us kernels per step from
    # Python / ctypes).  Every C-ABI entry point launches on the caller's stream, allocates nothing and keeps its scratch per
    # stream, so a whole step -- both passes of the coarse-to-fine loop, sampling, solve -- captures into ONE graph
    # (torch.cuda.graph: tensors the Python layer creates inside the capture live in the graph's private pool) and replays with a
    # single launch.  Seeds are kernel arguments: a replay repeats the draws of its capture (re-capture, or run eagerly, for fresh
    # draws per batch); timing hooks (ops.kernel_events / kernel_counters) must be off while capturing.
    def capture(self, seed=0, warmup=2, stream=None):
        """Capture step(seed) on `stream` (default: a new one); returns the static output tensors (H, good matches) every replay()
        refills.  (The HIP runtime deals a process's streams onto 4 hardware queues in the order of their first use -- rocprofv3's
        Queue_Id --, and two streams on one queue run one after the other: scenes that replay side by side want streams whose
        first uses were consecutive, which is why the stream can be handed in.)"""
        from gfnet_amd import ops

        if ops.kernel_events is not None or ops.kernel_counters is not None:
            raise RuntimeError("Scene.capture: switch ops.kernel_events / ops.kernel_counters off first (they record events / synchronise)")
        self._gstream = stream if stream is not None else torch.cuda.Stream()
        self._gstream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._gstream):
            for _ in range(warmup):  # lazy module state, the stream's scratch buffers, the allocator's pools
                self.step(seed)
        self._gstream.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph, stream=self._gstream):
            self._graph_out = self.step(seed)
        # (No wait_stream on the capture stream here: nothing has run yet, and on ROCm 7.2 an event recorded on the stream right
        # behind the end of its capture, waited for by the default stream, made a LATER replay fault once the default stream had
        # touched other tensors -- tools/dbg_graph.py, modes eager_inl / eager_inl_wait.)
        return self._graph_out

    def replay(self):
        """One captured step on the capture stream; returns the static outputs (valid once that stream has caught up)."""
        with torch.cuda.stream(self._gstream):
            self._graph.replay()
        return self._graph_out

    # The multi-stream arrangements of bench.py's SceneRunner as graphs: the stages of a step captured APART on a stream each --
    # stages=2: match | sampling + solve; stages=3: first pass | refinement pass + post-processing | sampling + solve --, twice each
    # (a stage graph's outputs are static buffers: the second copy lets step i + 1's stage run while step i's next stage still reads
    # the first), events between the replays.  A step's later stages then run beside the next steps' earlier ones, as in the eager
    # pipeline, at one launch per stage and step.
    def capture_pipelined(self, seed=0, warmup=2, streams=None, stages=2):
        from gfnet_amd import ops

        if ops.kernel_events is not None or ops.kernel_counters is not None:
            raise RuntimeError("Scene.capture_pipelined: switch ops.kernel_events / ops.kernel_counters off first")
        if stages == 2:
            fns = [lambda _: self.match(), lambda m: self.finish(m[0], m[1], seed)]
        else:
            fns = [lambda _: self.match_first(), lambda c: self.match_second(c), lambda m: self.finish(m[0], m[1], seed)]
        n = len(fns)
        cur = torch.cuda.current_stream()
        self._pst = list(streams) if streams is not None else [torch.cuda.Stream() for _ in range(n)]
        if len(self._pst) != n:
            raise ValueError(f"Scene.capture_pipelined: {n} stages need {n} streams")
        self._ms, self._fs = self._pst[0], self._pst[-1]
        x = None
        for st, fn in zip(self._pst, fns):  # lazy module state, every stream's scratch buffers, the allocator's pools
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                for _ in range(max(warmup, 1)):
                    y = fn(x)
            st.synchronize()
            x = y
        self._pgraph, self._pout = [[None, None] for _ in range(n)], [[None, None] for _ in range(n)]
        for k in range(2):
            x = None
            for j, (st, fn) in enumerate(zip(self._pst, fns)):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    x = fn(x)
                torch.cuda.synchronize()
                self._pgraph[j][k], self._pout[j][k] = g, x
        self._turn, self._pdone = 0, [[None, None] for _ in range(n)]
        return self._pout[-1]

    def replay_pipelined(self):
        """One step through the captured stages; returns this step's static outputs (H, good matches) and the event that says
        they are complete (they are overwritten by the replay after next)."""
        k = self._turn
        self._turn ^= 1
        n = len(self._pst)
        prev = None
        for j, st in enumerate(self._pst):
            if j + 1 < n and self._pdone[j + 1][k] is not None:
                st.wait_event(self._pdone[j + 1][k])  # the next stage of two steps ago has read this copy's outputs
            if prev is not None:
                st.wait_event(prev)
            with torch.cuda.stream(st):
                self._pgraph[j][k].replay()
            prev = st.record_event()
            self._pdone[j][k] = prev
        return self._pout[-1][k], prev

