"""Multi-process (world size 2, gloo, CPU) tests of the data-parallel glue around the hot path
(depth_estimation_amd/dist.py).  The path itself has no collective: what is checked is that the
batch shards partition the targets and that the scalar / gradient-bucket reductions give the single-process
answer."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _import_dist():
    # dist.py has no dependency on the HIP library; import it without triggering the package's ctypes load
    import importlib.util
    spec = importlib.util.spec_from_file_location("de_dist", os.path.join(ROOT, "depth_estimation_amd", "dist.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D = _import_dist()
        torch.manual_seed(0)  # same "global batch" on both ranks
        B = 5
        per_target_loss = torch.rand(B, dtype=torch.float32)
        per_target_log = torch.rand(B, dtype=torch.float32)
        lo, hi = D.shard_range(B, rank, world)
        mine_loss, mine_log = D.shard_batch([per_target_loss, per_target_log], rank, world)
        assert mine_loss.shape[0] == hi - lo
        partial = torch.stack([mine_loss.sum(), mine_log.sum(), torch.tensor(float(hi - lo))])
        total = D.reduce_scalars(partial)
        # gradient buckets: rank-dependent grads, two tiny buckets
        params = [torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(3, 5)), torch.nn.Parameter(torch.zeros(2))]
        for i, p in enumerate(params):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        gb = D.GradBuckets(params, bucket_bytes=64)
        gb.all_reduce()
        hs = gb_async = D.GradBuckets(params, bucket_bytes=1 << 20)
        for i, p in enumerate(params):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        handles = gb_async.all_reduce(async_op=True)
        for h in handles:
            h.wait()
        gb_async.unpack()
        q.put((rank, (lo, hi), total.tolist(), [float(p.grad.flatten()[0]) for p in params], len(gb.buckets),
               float(per_target_loss.sum()), float(per_target_log.sum())))
    finally:
        dist.destroy_process_group()


def _worker_uneven(rank, world, port, q):
    """Overlapped buckets where the ranks' gradients arrive in different orders and rank 1 leaves a whole bucket without
    a gradient (an unused branch): the collectives must still pair bucket with bucket."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D = _import_dist()
        # four parameters of EQUAL size, one per bucket: mispaired collectives would not hang, they would sum wrong buffers
        params = [torch.nn.Parameter(torch.full((8,), float(i + 1))) for i in range(4)]
        gb = D.GradBuckets(params, bucket_bytes=32, average=False)
        assert len(gb.buckets) == 4
        gb.attach()
        # a backward without begin() leaves the gradients local and launches nothing
        (params[0] * 1.0).sum().backward()
        assert float(params[0].grad[0]) == 1.0
        params[0].grad = None
        try:
            gb.finish()
            raised = False
        except RuntimeError:
            raised = True
        out = []
        for step in range(2):
            for p in params:
                p.grad = None
            gb.begin()
            if rank == 0:       # uses the parameters in the order 0,1,2,3: backward reaches 3 first
                loss = sum((i + 1) * (p * (10.0 + step)).sum() for i, p in enumerate(params))
            else:               # the other way round, and parameter 1 not at all
                loss = sum((i + 1) * (params[i] * (20.0 + step)).sum() for i in (3, 2, 0))
            loss.backward()
            gb.finish()
            out.append([float(p.grad[0]) for p in params])
        q.put((rank, raised, out))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_overlapped_buckets_fixed_order():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_uneven, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, raised, out in res:
        assert raised                                  # finish() without begin() is a clear error
        for step, got in enumerate(out):
            # SUM over ranks of (i+1) * factor; rank 1 contributes nothing to parameter 1
            want = [(i + 1) * ((10.0 + step) + (0.0 if i == 1 else 20.0 + step)) for i in range(4)]
            assert got == want, (rank, step, got, want)


def test_shard_range_partitions_every_batch():
    D = _import_dist()
    for B in (0, 1, 5, 12, 64):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi = D.shard_range(B, r, world)
                assert 0 <= lo <= hi <= B
                got += list(range(lo, hi))
            assert got == list(range(B))
            sizes = [D.shard_range(B, r, world)[1] - D.shard_range(B, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        D.shard_range(4, 2, 2)


def test_world2_gloo_scalar_and_bucket_reduction():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, t0, g0, nb0, full_loss, full_log), (r1, s1, t1, g1, nb1, _, _) = res
    assert s0 == (0, 3) and s1 == (3, 5)
    assert t0 == t1                                  # every rank holds the same reduced scalars
    assert abs(t0[0] - full_loss) < 1e-5 and abs(t0[1] - full_log) < 1e-5 and t0[2] == 5.0
    # mean over ranks of (rank+1)*(i+1) = 1.5*(i+1)
    assert g0 == g1 == [1.5, 3.0, 4.5]
    assert nb0 >= 2                                   # 64-byte buckets split the 3 parameters


def test_reduce_scalars_is_identity_without_process_group():
    D = _import_dist()
    v = torch.tensor([1.0, 2.0, 3.0])
    assert torch.equal(D.reduce_scalars(v), v)
