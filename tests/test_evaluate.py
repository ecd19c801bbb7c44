"""Evaluation harness (gfnet_amd/evaluate.py): dataset layout of the reference (README.md:47-53,
test.py:61-64), batched match -> sample -> solve, AUC/ACE bookkeeping, error gathering across ranks."""
import json
import os
import socket

import numpy as np
import pytest
import torch


def _write_dataset(root, homographies, sizes, ext="png"):
    from PIL import Image

    for d in ("source", "target", "H_s2t"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    rng = np.random.default_rng(3)
    for i, (H, (w, h)) in enumerate(zip(homographies, sizes)):
        name = f"pair_{i:03d}"
        for d in ("source", "target"):
            Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(os.path.join(root, d, f"{name}.{ext}"))
        with open(os.path.join(root, "H_s2t", name + ".json"), "w") as f:
            json.dump({"H": np.asarray(H).tolist()}, f)


def _homographies(n, size, seed=5, corner=0.12):
    """Random 4-corner perturbations (datasets/generate_random_H_large_size.py style), as 3x3 matrices."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        src = np.array([[0, 0], [size - 1, 0], [size - 1, size - 1], [0, size - 1]], np.float64)
        dst = src + rng.uniform(-corner, corner, (4, 2)) * size
        A = []
        for (x, y), (u, v) in zip(src, dst):
            A.append([x, y, 1, 0, 0, 0, -u * x, -u * y, -u])
            A.append([0, 0, 0, x, y, 1, -v * x, -v * y, -v])
        h = np.linalg.svd(np.asarray(A))[2][-1]
        out.append((h / h[8]).reshape(3, 3))
    return out


def test_list_pairs_follows_the_reference_layout(tmp_path):
    from gfnet_amd import evaluate

    Hs = _homographies(3, 32)
    _write_dataset(str(tmp_path), Hs, [(32, 32)] * 3, ext="jpg")
    pairs = evaluate.list_pairs(str(tmp_path))
    assert [os.path.basename(p[0]) for p in pairs] == ["pair_000.jpg", "pair_001.jpg", "pair_002.jpg"]
    assert all(p[1].endswith(os.path.join("target", os.path.basename(p[0]))) for p in pairs)
    assert pairs[1][2].endswith(os.path.join("H_s2t", "pair_001.json"))
    np.testing.assert_allclose(evaluate.load_homography(pairs[2][2]), np.asarray(Hs[2], np.float32))
    assert evaluate.list_pairs(str(tmp_path), ext="png") == []
    os.remove(pairs[0][2])
    with pytest.raises(FileNotFoundError):
        evaluate.list_pairs(str(tmp_path))
    with pytest.raises(FileNotFoundError):
        evaluate.list_pairs(str(tmp_path / "nowhere"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gather_worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from gfnet_amd import evaluate, parallel

    parallel.init_from_env(backend="gloo")
    lo, hi = parallel.shard_range(n, rank, world)
    errs, t, w = evaluate._gather(np.arange(lo, hi, dtype=np.float64) * 0.5, 1.0 + rank, 3.0 + rank, n, lo, world)
    q.put((rank, errs, (t, w)))
    dist.barrier()
    dist.destroy_process_group()


def test_error_gather_world2_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, 7, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, errs, t in res:
        np.testing.assert_array_equal(errs, np.arange(7) * 0.5)
        assert t[0] == pytest.approx(1.5) and t[1] == pytest.approx(3.5)


class _OracleMatcher:
    """Stands in for a trained GFNet: returns the dense warp of the ground-truth homography (what match() converges
    to), so that everything after match -- sampling, coordinate conventions, solve, errors -- is what is tested."""
    sample_mode = "threshold_balanced"
    sample_thresh = 0.05

    def __init__(self, Hs, G=64):
        self.Hs, self.G, self.calls = Hs, G, []

    def match_batch(self, A, B):
        n, _, h, w = A.shape
        k0 = sum(self.calls)
        self.calls.append(n)
        G = self.G
        lin = (torch.arange(G, dtype=torch.float64) * 2 + 1) / G - 1
        gy, gx = torch.meshgrid(lin, lin, indexing="ij")
        px, py = (w - 1) * (gx + 1) / 2, (h - 1) * (gy + 1) / 2
        warps = []
        for k in range(n):
            H = torch.from_numpy(np.asarray(self.Hs[k0 + k], np.float64))
            den = H[2, 0] * px + H[2, 1] * py + H[2, 2]
            u = (H[0, 0] * px + H[0, 1] * py + H[0, 2]) / den
            v = (H[1, 0] * px + H[1, 1] * py + H[1, 2]) / den
            warps.append(torch.stack((gx, gy, 2 * u / (w - 1) - 1, 2 * v / (h - 1) - 1), -1))
        warp = torch.stack(warps).float().cuda()
        inside = ((warp[..., 2:].abs() <= 1).all(-1)).float()
        return warp, inside * 0.9 + 0.01


@pytest.mark.gpu
def test_evaluate_recovers_known_homographies(tmp_path):
    from gfnet_amd import evaluate

    sizes = [(96, 96)] * 5 + [(80, 64)] * 2          # a size change closes a batch
    Hs = _homographies(7, 64, corner=0.08)
    _write_dataset(str(tmp_path), Hs, sizes)
    m = _OracleMatcher(Hs)
    seen = []
    res = evaluate.evaluate(m, str(tmp_path), batch_size=4, num_samples=2000, progress=lambda i, n: seen.append((i, n)))
    assert m.calls == [4, 1, 2]
    assert seen[-1] == (7, 7) and res["n"] == 7 and res["errors"].shape == (7,)
    assert np.all(res["errors"] < 1e-2), res["errors"]   # px: exact correspondences, fp32 coordinates
    assert res["ace"] == pytest.approx(float(np.mean(res["errors"])))
    assert all(res[f"auc@{t}"] > 0.99 for t in (3, 5, 10, 20))
    assert res["time"] > 0


@pytest.mark.gpu
def test_evaluate_time_is_device_time_under_a_slow_loader(tmp_path, monkeypatch):
    """with a loader slower than the device work (real JPEG / PNG sets), `time` must still be the device seconds per pair --
    every batch's [first upload, solved] interval on the HIP-event clock, merged -- not wall time minus decoding (which also took away
    the device work that ran under the decoding of the next batch), and `wall_time` carries the decoding."""
    import time as _time

    from gfnet_amd import evaluate

    sizes = [(96, 96)] * 6
    Hs = _homographies(6, 64, corner=0.08)
    _write_dataset(str(tmp_path), Hs, sizes)

    class Busy(_OracleMatcher):  # ~20 ms of device work per batch that no host timer inside evaluate() sees
        def match_batch(self, A, B):
            torch.cuda._sleep(40_000_000)
            return super().match_batch(A, B)

    real = evaluate._load_image
    monkeypatch.setattr(evaluate, "_load_image", lambda p: (_time.sleep(0.03), real(p))[1])  # 60 ms of "decoding" per pair
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); torch.cuda._sleep(40_000_000); ev1.record(); torch.cuda.synchronize()
    sleep_s = ev0.elapsed_time(ev1) * 1e-3
    res = evaluate.evaluate(Busy(Hs), str(tmp_path), batch_size=2, num_samples=2000)
    per_pair_device = 3 * sleep_s / 6                     # three batches, each at least one sleep kernel long
    assert res["time"] >= 0.9 * per_pair_device, (res["time"], per_pair_device)
    assert res["wall_time"] >= 0.06 and res["wall_time"] >= res["time"]
    assert res["time"] < res["wall_time"]                 # decoding is in wall_time only


class _TwoCallMatcher(_OracleMatcher):
    """The same matcher with the matching offered in two calls (GFNet.match_batch_first / match_batch_second): evaluate() then
    streams batches through three HIP streams."""

    def match_batch_first(self, A, B):
        self.first_streams = getattr(self, "first_streams", []) + [torch.cuda.current_stream().cuda_stream]
        return {"A": A, "B": B, "nested": [A.new_zeros(3), (B.new_ones(2),)]}

    def match_batch_second(self, state):
        self.second_streams = getattr(self, "second_streams", []) + [torch.cuda.current_stream().cuda_stream]
        return self.match_batch(state["A"], state["B"])


@pytest.mark.gpu
def test_evaluate_streams_a_two_call_matcher_through_three_streams(tmp_path):
    from gfnet_amd import evaluate

    sizes = [(96, 96)] * 6
    Hs = _homographies(6, 64, corner=0.08)
    _write_dataset(str(tmp_path), Hs, sizes)
    torch.manual_seed(5)  # the sampler's seeds come from torch's generator
    ref = evaluate.evaluate(_OracleMatcher(Hs), str(tmp_path), batch_size=2, num_samples=2000)
    m = _TwoCallMatcher(Hs)
    torch.manual_seed(5)
    res = evaluate.evaluate(m, str(tmp_path), batch_size=2, num_samples=2000)
    assert m.calls == [2, 2, 2]
    assert len(set(m.first_streams)) == 1 and len(set(m.second_streams)) == 1 and m.first_streams[0] != m.second_streams[0]
    np.testing.assert_array_equal(res["errors"], ref["errors"])  # same matches, same seeds: the same homographies
    assert np.all(res["errors"] < 1e-2)


@pytest.mark.gpu
def test_match_batch_resizes_and_calls_the_backbone():
    """GFNet.match_batch: resize + normalise per pass (448 bicubic, 560 bilinear), backbone on cat(A, B)."""
    from gfnet_amd.model.network import GFNet
    from gfnet_amd import ops
    import oracle

    calls = []

    class Stop(Exception):
        pass

    def backbone(x, upsample):
        calls.append((tuple(x.shape), bool(upsample), x.clone()))
        raise Stop

    conf = {"matcher": {"num_grid": [32, 32, 64, 128, 256], "radius": [7, 6, 4, 2, 0], "num_itr": [1, 1, 1, 1, 1],
                        "displacement_dim": [64, 64, 32, 16, 8]}, "encoder_cfg": {"feat_chs": [64, 32, 16, 8]}}
    model = GFNet(conf, symmetric=True, upsample_preds=True, attenuate_cert=True, backbone=backbone).cuda().eval()
    a = torch.rand(2, 3, 50, 70)
    b = torch.rand(2, 3, 50, 70)
    with pytest.raises(Stop):
        model.match_batch(a, b)
    shape, up, x = calls[0]
    assert shape == (4, 3, 448, 448) and up is False
    want = oracle.resize_normalise(torch.cat((a, b)).numpy(), (448, 448), "bicubic")
    np.testing.assert_allclose(x.cpu().numpy(), want, atol=2e-5)


@pytest.mark.gpu
def test_match_batch_in_two_calls_equals_match_batch():
    """GFNet.match_batch_first + match_batch_second (what evaluate() streams through three HIP streams) return exactly
    match_batch's warp and certainty -- toy backbone (pooled image channels), the refiners' real conv stacks, both passes."""
    import math

    import torch.nn.functional as F
    from gfnet_amd.model.network import GFNet

    chs = {"16": 64, "8": 64, "4": 32, "2": 16, "1": 8}

    def backbone(x, upsample):
        pyr = {}
        for s in (["8", "4", "2", "1"] if upsample else ["16", "8", "4", "2", "1"]):
            k = int(s)
            f = F.avg_pool2d(x, k) if k > 1 else x
            if s == "16":  # the coarsest features come from the patch-14 trunk: 32 x 32 at 448 (network.py:329's grid rule)
                f = F.adaptive_avg_pool2d(x, (x.shape[-2] // 14, x.shape[-1] // 14))
            f = f.repeat(1, math.ceil(chs[s] / 3), 1, 1)[:, :chs[s]]
            f = f * torch.linspace(0.5, 1.5, chs[s], device=x.device).view(1, -1, 1, 1)
            pyr[s] = f.contiguous()
        n = x.shape[0] // 2
        return {s: f[:n] for s, f in pyr.items()}, {s: f[n:] for s, f in pyr.items()}

    conf = {"matcher": {"num_grid": [32, 32, 64, 128, 256], "radius": [7, 6, 4, 2, 0], "num_itr": [1, 1, 1, 1, 1],
                        "displacement_dim": [64, 64, 32, 16, 8]}, "encoder_cfg": {"feat_chs": [64, 32, 16, 8]}}
    torch.manual_seed(0)
    model = GFNet(conf, symmetric=True, upsample_preds=True, attenuate_cert=True, backbone=backbone).cuda().eval()
    a, b = torch.rand(1, 3, 60, 80), torch.rand(1, 3, 60, 80)
    warp, cert = model.match_batch(a, b)
    warp2, cert2 = model.match_batch_second(model.match_batch_first(a, b))
    assert torch.isfinite(warp).all() and torch.equal(warp, warp2) and torch.equal(cert, cert2)
