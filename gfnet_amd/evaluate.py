"""Evaluation harness -- the counterpart of the reference's test.py:57-75 loop and of
benchmark/multimodal_homog_benchmark_multiscale.py:54-78, batched and sharded.

Dataset layout (README.md:47-53, test.py:61-64, datasets/homography_dataset_large_size.py:113-119):

    <root>/source/<name>.<ext>     source image
    <root>/target/<name>.<ext>     target image (same file name)
    <root>/H_s2t/<name>.json       {"H": [[..3x3..]]}  ground-truth homography source -> target

The reference walks the pairs one by one (match -> sample -> cv2.findHomography -> corner error) and prints
AUC@3/5/10/20, ACE and the mean runtime.  Here pairs are decoded on the host, grouped into batches of equal image
size, and every batch runs match -> sample -> solve on the device in one go; with torch.distributed initialised the
pairs are sharded over the ranks (parallel.shard_range) and the per-pair errors gathered at the end.

The matcher is any object with the reference's surface:
    match(im_a, im_b)                         # per pair (PIL images), as test.py uses it, or, preferred,
    match_batch(im_a, im_b)                   # (B,3,H,W) float tensors in [0,1] on the GPU -> (warp, certainty)
plus `sample` semantics through gfnet_amd.model.network.sample_batched.
"""
import json
import os
import time

import numpy as np
import torch

from . import parallel
from .estimation import auc, corner_error, estimate_homographies
from .utils.image import to_tensor

THRESHOLDS = (3, 5, 10, 20)


def list_pairs(root, ext=None):
    """[(source path, target path, H json path)] in sorted order.  `root` is the directory that holds source/,
    target/ and H_s2t/ (test.py:61-64 derives the other two from the source path the same way)."""
    src = os.path.join(root, "source")
    if not os.path.isdir(src):
        raise FileNotFoundError(f"{src}: expected <root>/source, <root>/target, <root>/H_s2t")
    pairs = []
    for name in sorted(os.listdir(src)):
        stem, e = os.path.splitext(name)
        if ext is not None and e.lstrip(".").lower() != ext.lower():
            continue
        tgt = os.path.join(root, "target", name)
        hj = os.path.join(root, "H_s2t", stem + ".json")
        if not os.path.isfile(tgt) or not os.path.isfile(hj):
            raise FileNotFoundError(f"pair '{name}': missing {tgt if not os.path.isfile(tgt) else hj}")
        pairs.append((os.path.join(src, name), tgt, hj))
    return pairs


def load_homography(path):
    """H_s2t as float32, like estimation.py:52-53."""
    with open(path, "r") as f:
        return np.array(json.load(f)["H"], dtype=np.float32)


def _load_image(path):
    from PIL import Image

    return to_tensor(Image.open(path).convert("RGB"))


def evaluate(matcher, root, batch_size=32, num_samples=5000, thresholds=THRESHOLDS, ext=None, seed=0, progress=None):
    """Run the whole test set.  Returns a dict: auc@t, ace (mean corner error), time, wall_time, errors (per pair, dataset order), n.
    `time` = seconds of DEVICE work per pair: for every batch a HIP event pair from its first upload to its solved matrices, the
    batches' busy intervals merged (they overlap across the streams) -- what the reference brackets per pair, estimation.py:56-78:
    match + sample + solve, host decoding excluded.  `wall_time` = wall seconds of the whole loop per pair, decoding included.
    (Round 4 reported wall minus decoding, which also subtracted device work hidden under the decoding of the next batch: ADVICE r4.)
    Every rank returns the full result."""
    from .model.network import sample_batched

    pairs = list_pairs(root, ext)
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    rank = torch.distributed.get_rank() if dist_on else 0
    world = torch.distributed.get_world_size() if dist_on else 1
    lo, hi = parallel.shard_range(len(pairs), rank, world)
    mine = pairs[lo:hi]
    errors = np.full(len(mine), np.nan, np.float64)
    # Batches stream through two (or three, below) HIP streams: the matching of batch k + 1 (chip-wide launches) runs while batch k
    # is sampled and solved on the second stream (a third of that stage is one-workgroup-per-pair kernels -- curve sort, radix select, LM finish --
    # that leave most of the chip idle), and batch k's 3x3 matrices come back through pinned memory behind an event instead of a
    # blocking copy.  bench.py times the same arrangement (`config.step_pipeline`).
    # A matcher that offers the matching in two calls (GFNet.match_batch_first / match_batch_second: first pass | refinement pass)
    # gets a third stream: batch k + 1's first pass beside batch k's refinement pass (bench.py: 10.6 k -> 11.3 k pairs/s).
    three = hasattr(matcher, "match_batch_first") and hasattr(matcher, "match_batch_second")
    pool = parallel.concurrent_streams(3 if three else 2)  # streams tested to sit on different hardware queues (parallel.py)
    m1, ms, fs = (pool[0], pool[1], pool[2]) if three else (None, pool[0], pool[1])
    pending = None  # (first index, n, ground-truth Hs, sizes, pinned H, event)

    def hand_over(obj, stream):  # tensors produced on one stream, consumed on another: tell the caching allocator
        if torch.is_tensor(obj):
            if obj.is_cuda:
                obj.record_stream(stream)
        elif isinstance(obj, dict):
            for v in obj.values():
                hand_over(v, stream)
        elif isinstance(obj, (list, tuple)):
            for v in obj:
                hand_over(v, stream)

    def settle(p):
        first, n_, Hs_, (w1_, h1_), Hpin, ev = p
        ev.synchronize()
        Hp = Hpin.numpy()
        for k in range(n_):
            errors[first + k] = corner_error(Hs_[k], Hp[k], w1_, h1_)

    i = 0
    t_loop = time.perf_counter()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev0.record(ms)
    spans = []  # per batch: (event at its first upload, event behind its solve)
    while i < len(mine):
        # a batch = consecutive pairs whose images have the same size (test sets are uniform; a change closes the batch)
        ims_a, ims_b, Hs = [], [], []
        while i + len(ims_a) < len(mine) and len(ims_a) < batch_size:
            a, b, hj = mine[i + len(ims_a)]
            ta, tb = _load_image(a), _load_image(b)
            if ims_a and (ta.shape != ims_a[0].shape or tb.shape != ims_b[0].shape):
                break
            ims_a.append(ta)
            ims_b.append(tb)
            Hs.append(load_homography(hj))
        n = len(ims_a)
        h1, w1 = ims_a[0].shape[-2:]
        h2, w2 = ims_b[0].shape[-2:]
        sa, sb = torch.stack(ims_a), torch.stack(ims_b)
        began = torch.cuda.Event(enable_timing=True)
        with torch.inference_mode():
            if three:
                with torch.cuda.stream(m1):
                    began.record(m1)
                    A, Bt = sa.cuda(non_blocking=True), sb.cuda(non_blocking=True)
                    state = matcher.match_batch_first(A, Bt)
                    first_done = m1.record_event()
                with torch.cuda.stream(ms):
                    ms.wait_event(first_done)
                    hand_over(state, ms)
                    warp, cert = matcher.match_batch_second(state)
                    good = None
                    matched = ms.record_event()
            else:
                with torch.cuda.stream(ms):
                    began.record(ms)
                    A, Bt = sa.cuda(non_blocking=True), sb.cuda(non_blocking=True)
                    if hasattr(matcher, "match_batch"):
                        warp, cert = matcher.match_batch(A, Bt)
                        good = None
                    else:  # the reference's per-pair surface
                        gs = []
                        for k in range(n):
                            w_, c_ = matcher.match(A[k:k + 1], Bt[k:k + 1])
                            gs.append(matcher.sample(w_, c_, num_samples)[0])
                        good = torch.stack(gs)
                        warp = cert = None
                    matched = ms.record_event()
            with torch.cuda.stream(fs):
                fs.wait_event(matched)
                for t in (warp, cert, good):
                    if t is not None:
                        t.record_stream(fs)
                if good is None:
                    good, _ = sample_batched(matcher, warp, cert, num_samples)
                Hdev = estimate_homographies(good, (w1, h1, w2, h2), seed=seed + lo + i)
                Hpin = torch.empty(Hdev.shape, dtype=Hdev.dtype, pin_memory=True)
                Hpin.copy_(Hdev, non_blocking=True)   # the only device->host copy of a batch
                solved = torch.cuda.Event(enable_timing=True)
                solved.record(fs)
                spans.append((began, solved))
        if pending is not None:
            settle(pending)  # batch k - 1: its matrices have had the whole matching of batch k to arrive
        pending = (i, n, Hs, (w1, h1), Hpin, solved)
        i += n
        if progress:
            progress(lo + i, len(pairs))
    if pending is not None:
        settle(pending)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_loop
    # device-busy seconds: union of the batches' [first upload, solved] intervals on the common clock of the HIP events
    iv = sorted((ev0.elapsed_time(a), ev0.elapsed_time(b)) for a, b in spans)
    busy, end = 0.0, float("-inf")
    for a, b in iv:
        if a > end:
            busy += b - a
            end = b
        elif b > end:
            busy += b - end
            end = b
    elapsed = busy * 1e-3
    all_err, total_time, total_wall = _gather(errors, elapsed, wall, len(pairs), lo, world)
    res = {f"auc@{t}": v for t, v in zip(thresholds, auc(all_err, thresholds))}
    res.update(ace=float(np.mean(all_err)), time=total_time / max(len(pairs), 1), wall_time=total_wall / max(len(pairs), 1),
               errors=all_err, n=len(pairs))
    return res


def _gather(errors, elapsed, wall, n_total, lo, world):
    if world == 1:
        return errors, elapsed, wall
    import torch.distributed as dist

    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    buf = torch.zeros(n_total + 2, dtype=torch.float64, device=dev)
    buf[lo:lo + len(errors)] = torch.from_numpy(errors).to(dev)
    buf[n_total] = elapsed
    buf[n_total + 1] = wall
    dist.all_reduce(buf)  # disjoint slots: the sum is the concatenation; the last two slots sum the ranks' device and wall time
    out = buf.cpu().numpy()
    return out[:n_total], float(out[n_total]) / world, float(out[n_total + 1]) / world  # ranks run concurrently: ~ the mean of the ranks


def main(argv=None):
    """python -m gfnet_amd.evaluate --conf_path gfnet_configs/basic.json --root <dataset dir> --backbone module:factory
    The backbone factory returns a callable (images (2B,3,H,W), upsample) -> (pyramid A, pyramid B) -- e.g. a wrapper
    around the reference's GFNet.extract_features; weights are not part of this package."""
    import argparse
    import importlib

    from .model.network import GFNet

    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--conf_path", required=True)
    ap.add_argument("--root", required=True, help="directory with source/ target/ H_s2t/")
    ap.add_argument("--backbone", required=True, help="module:function returning the backbone callable")
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--ext", default=None)
    ap.add_argument("--conv_precision", choices=("fp32", "fp16", "amp"), default="fp32")
    ap.add_argument("--refiner_ckpt", default=None, help="torch checkpoint with a 'model' state_dict (conv_refiner.* entries are loaded)")
    args = ap.parse_args(argv)
    rank, world, _ = parallel.init_from_env()
    with open(args.conf_path) as f:
        conf = json.load(f)
    mod, fn = args.backbone.split(":")
    backbone = getattr(importlib.import_module(mod), fn)()
    model = GFNet(conf, initial_res=(448, 448), upsample_res=(560, 560), symmetric=True, upsample_preds=True, attenuate_cert=True,
                  backbone=backbone).cuda().eval()
    if args.refiner_ckpt:
        sd = torch.load(args.refiner_ckpt, map_location="cuda")["model"]
        model.conv_refiner.load_state_dict({k[len("conv_refiner."):]: v for k, v in sd.items() if k.startswith("conv_refiner.")})
    for r in model.conv_refiner.values():
        r.conv_precision = args.conv_precision
    res = evaluate(model, args.root, batch_size=args.batch_size, ext=args.ext)
    if rank == 0:
        name = os.path.basename(os.path.normpath(args.root))
        print({f"{k}_{name}": v for k, v in res.items() if k.startswith("auc@")})
        print(f"ACE: {res['ace']}")
        print(f"Time: {res['time']}")  # device seconds per pair (the reference's bracket: match + sample + solve)
        print(f"Wall time: {res['wall_time']}")
    return res


if __name__ == "__main__":
    main()
