"""World-size-2 gloo test of the only collective on the path: sharding pairs and gathering H."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pairs, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from gfnet_amd import parallel

    r, w, _ = parallel.init_from_env(backend="gloo")
    lo, hi = parallel.shard_range(n_pairs, r, w)
    # every pair's "H" is a function of its global index, so the gathered order can be checked
    H_local = torch.stack([torch.full((3, 3), float(i), dtype=torch.float64) + torch.eye(3, dtype=torch.float64)
                           for i in range(lo, hi)]) if hi > lo else torch.zeros((0, 3, 3), dtype=torch.float64)
    counts = [parallel.shard_range(n_pairs, k, w)[1] - parallel.shard_range(n_pairs, k, w)[0] for k in range(w)]
    H_all = parallel.gather_homographies(H_local, counts)
    q.put((rank, lo, hi, H_all.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _run(n_pairs, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res)


def test_shard_range_partitions_exactly():
    from gfnet_amd.parallel import shard_range

    for n in (0, 1, 7, 32, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_gather_homographies_world2_even_and_ragged():
    for n_pairs in (8, 7):
        res = _run(n_pairs)
        expect = np.stack([np.full((3, 3), float(i)) + np.eye(3) for i in range(n_pairs)])
        for rank, lo, hi, H_all in res:
            np.testing.assert_array_equal(H_all, expect)
        assert res[0][1] == 0 and res[-1][2] == n_pairs


def test_single_process_is_a_no_op():
    from gfnet_amd import parallel

    H = torch.eye(3, dtype=torch.float64)[None]
    assert parallel.gather_homographies(H) is H


def _bench(*argv, env=None):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), env=e, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r.returncode, [json.loads(ln) for ln in lines], r.stderr


def test_bench_starts_its_own_ranks_when_typed_as_a_bare_command():
    """`python bench.py --gpus 2` without torchrun (WORLD_SIZE unset): the parent starts two rank processes, gloo stands in for
    RCCL, no kernels run (--dry-run); exactly ONE JSON line comes back, from rank 0, and it saw both ranks."""
    rc, js, err = _bench("--gpus", "2", "--backend", "gloo", "--dry-run")
    assert rc == 0, err
    assert len(js) == 1
    assert js[0]["n_gpus"] == 2 and js[0]["n_ranks_seen"] == 2 and js[0]["gather_ok"] is True


def test_bench_under_a_launcher_does_not_relaunch():
    """With RANK / WORLD_SIZE already in the environment (torch.distributed.run) bench.py is a rank, not a launcher: a
    world of one given by the environment runs in place."""
    rc, js, err = _bench("--gpus", "1", "--dry-run", env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert rc == 0, err
    assert len(js) == 1 and js[0]["n_gpus"] == 1


def test_bench_child_failure_is_the_parents_return_code():
    """--gpus 2 given to ranks whose environment says WORLD_SIZE=1 must fail loudly (a mismatch is never papered over)."""
    rc, js, err = _bench("--gpus", "2", "--dry-run", env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert rc != 0 and not js


def test_bench_a_rank_dying_at_start_up_ends_the_run_at_once():
    """Rank 1 exits with code 3 before the rendezvous: the parent must terminate rank 0 (which would otherwise wait in the
    rendezvous until the process-group timeout) and hand that code back, within seconds."""
    import time

    t0 = time.time()
    rc, js, err = _bench("--gpus", "2", "--backend", "gloo", "--dry-run", env={"GFN_BENCH_TEST_EXIT_RANK": "1"})
    assert rc == 3 and not js, (rc, err[-500:])
    assert time.time() - t0 < 120
