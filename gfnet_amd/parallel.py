"""Multi-GPU layout of the path: image pairs shard embarrassingly, one process per GPU
(torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" for the CPU tests).  The reference has
no inference-time parallelism at all (scripts/test_script.sh pins one GPU, test.py:61 loops pairs
one by one); there is no exchange step anywhere on the path, so the only collective is the gather of
the estimated 3x3 matrices (72 bytes per pair) at the end of a batch.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun); no-op for one process.
    Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            # bind the communicator to this rank's GPU: without device_id c10d guesses the device from the global rank at the
            # first collective ("this can cause a hang if the mapping is incorrect")
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_range(n_items, rank, world):
    """Contiguous block of items for `rank`: sizes differ by at most one, earlier ranks get the extra."""
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def gather_homographies(H_local, counts=None):
    """All ranks end up with every pair's H, in pair order.  H_local: (n_local,3,3).  With unequal
    shard sizes pass counts (list of per-rank sizes); blocks are padded to the largest for the
    collective and trimmed afterwards."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return H_local
    world = dist.get_world_size()
    if counts is None:
        counts = [H_local.shape[0]] * world
    nmax = max(counts)
    buf = H_local.new_zeros((nmax, 3, 3))
    buf[: H_local.shape[0]] = H_local
    out = H_local.new_empty((world * nmax, 3, 3))
    dist.all_gather_into_tensor(out, buf.contiguous())
    out = out.view(world, nmax, 3, 3)
    return torch.cat([out[r, : counts[r]] for r in range(world)], dim=0)


# ---- streams that really run side by side -------------------------------------------------------------------------------------------
# The HIP runtime deals a process's streams onto a few hardware queues (4 by default, GPU_MAX_HW_QUEUES) as they are first used, not
# one to one: on ROCm 7.2 the fourth and fifth stream of a process share a queue, and two streams on one queue run one after the other
# whatever the events between them say (rocprofv3's Queue_Id).  The two-stream pipelines (match | sampling + solve) and the
# scene-per-stream steps therefore take their streams from here: a process-wide pool whose members were TESTED to overlap -- two
# one-workgroup spin kernels, one per stream, take as long as one when the queues differ and as long as two when they are shared.
_stream_pool = {}


def _overlap(a, b, cycles=400_000, trials=3):
    """True when work on streams a and b runs concurrently (a one-workgroup spin on each: ~0.2 ms alone): the MEDIAN verdict of
    `trials` measurements (one 0.2-ms wall-time comparison misjudges a pair on a busy GPU or across a clock change)."""
    votes = sorted(_overlap_ratio(a, b, cycles) for _ in range(trials))
    return votes[len(votes) // 2] < 1.5


def _overlap_ratio(a, b, cycles):
    """(time of one spin on each stream, started together) / (time of one spin alone): ~1 side by side, ~2 on a shared queue."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def spin_alone(st):
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            ev[0].record()
            torch.cuda._sleep(cycles)
            ev[1].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1])

    spin_alone(a)  # clocks up, both streams' queues created: the reference time below is taken warm
    spin_alone(b)
    alone = min(spin_alone(a), spin_alone(b))
    start = torch.cuda.Event()
    start.record()          # both streams wait for the same point, then spin
    a.wait_event(start)
    b.wait_event(start)
    with torch.cuda.stream(a):
        ev[0].record()
        torch.cuda._sleep(cycles)
        ev[1].record()
    with torch.cuda.stream(b):
        ev[2].record()
        torch.cuda._sleep(cycles)
        ev[3].record()
    torch.cuda.synchronize()
    both = max(ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3]), ev[0].elapsed_time(ev[3]))
    return both / max(alone, 1e-6)


def concurrent_streams(n, device=None, tries=16):
    """n HIP streams of the current device that pairwise run concurrently (see above); the same streams on every call (a pool that
    grows as needed).  Falls back to plain new streams for members it cannot find in `tries` candidates (more streams than
    hardware queues): the work is still correct, only less overlapped."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    pool = _stream_pool.setdefault(dev, [])
    with torch.cuda.device(dev):
        rejected = 0
        while len(pool) < n and rejected < tries:
            cand = torch.cuda.Stream()
            if all(_overlap(cand, s) for s in pool):
                pool.append(cand)
            else:
                rejected += 1
        if len(pool) < n:
            import warnings

            warnings.warn(f"gfnet_amd.parallel: {n} streams asked for, {len(pool)} found that run side by side after {rejected} rejected "
                          "candidates; the rest are untested streams (correct, possibly serialised with another member)", RuntimeWarning)
        while len(pool) < n:
            pool.append(torch.cuda.Stream())
    return pool[:n]


def release_streams(device=None):
    """Drop the pool of a device (the streams are pinned for the life of the process otherwise)."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    _stream_pool.pop(dev, None)
