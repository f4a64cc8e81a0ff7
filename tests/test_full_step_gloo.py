"""BASELINE config 5 harness (scripts/full_step.py) on the CPU: world size 2, gloo.  The hot-path functions are replaced by
small differentiable torch stand-ins (the product path is HIP-only and has no CPU form); what is checked is the harness
itself -- stand-in networks with exact parameter counts, gradient buckets all-reduced from backward hooks, clipping and
Adam (train.py:102-109): two ranks with one sample each must end the step with the parameters of one process that
trained on both samples (the samples of a batch are summed into the loss, train.py:88, so the ranks' gradients are too)."""
import importlib.util
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTS = dict(pose=40_000, decoder=3_001, convgru=777)      # small, odd: the filler must make them exact
H, W, S = 16, 24, 2


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _torch_path():
    """differentiable torch stand-ins for the three hot-path functions (scripts/full_step.torch_stand_in_path)"""
    return _load("full_step", "scripts/full_step.py").torch_stand_in_path()


def _setup(fs, D, average=False):
    nets = fs.build_networks("cpu", COUNTS, seed=3)
    params = fs.trainable(nets)
    opt = torch.optim.Adam(params, lr=1e-2)
    buckets = D.GradBuckets(params, bucket_bytes=16 << 10, average=average)   # 16 KB: several buckets
    return nets, params, opt, buckets


def _batch():
    g = torch.Generator().manual_seed(5)
    return torch.rand(2, S + 1, 3, H, W, generator=g)          # the global batch: two samples


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        fs, D = _load("full_step", "scripts/full_step.py"), _load("de_dist", "depth_estimation_amd/dist.py")
        nets, params, opt, buckets = _setup(fs, D)
        buckets.attach()
        imgs = _batch()[rank:rank + 1]                          # this rank's sample
        for _ in range(2):
            loss = fs.train_step(nets, opt, buckets, imgs, _torch_path())
        grads = torch.cat([p.grad.flatten() for p in params])
        flat = torch.cat([p.detach().flatten() for p in params])
        q.put((rank, float(loss), flat.numpy(), grads.numpy(), len(buckets.buckets), buckets.nbytes()))
    finally:
        dist.destroy_process_group()


def test_stand_in_networks_have_the_reference_parameter_counts():
    fs = _load("full_step", "scripts/full_step.py")
    assert fs.REF_PARAMS == dict(pose=11_182_662, decoder=25_908_589, convgru=491_577)
    assert sum(fs.REF_PARAMS.values()) * 4 == 150_331_312          # the 150.3 MB of SURVEY 5
    nets = fs.build_networks("cpu", COUNTS)
    for name in COUNTS:
        assert sum(p.numel() for p in getattr(nets, name).parameters()) == COUNTS[name]
    # the full-size modules: count only (meta device, nothing allocated)
    with torch.device("meta"):
        big = dict(pose=fs.PoseStandIn(), decoder=fs.DecoderStandIn(), convgru=fs.ConvGruStandIn())
    for name, m in big.items():
        assert sum(p.numel() for p in m.parameters()) == fs.REF_PARAMS[name]


def test_world2_sharded_step_equals_single_process_step():
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, both samples, no collective
    torch.set_num_threads(1)
    fs, D = _load("full_step", "scripts/full_step.py"), _load("de_dist", "depth_estimation_amd/dist.py")
    nets, params, opt, _ = _setup(fs, D)
    for _ in range(2):
        loss = fs.train_step(nets, opt, None, _batch(), _torch_path())
    want_p = torch.cat([p.detach().flatten() for p in params]).numpy()
    want_g = torch.cat([p.grad.flatten() for p in params]).numpy()
    (r0, l0, p0, g0, nb, nbytes), (r1, l1, p1, g1, _, _) = res
    assert nb >= 3 and nbytes == 4 * sum(COUNTS.values())
    assert (p0 == p1).all() and (g0 == g1).all()                 # replicas stay bit-identical
    assert abs((l0 + l1) - float(loss)) < 1e-5 * abs(float(loss))   # the ranks' losses add up to the batch loss
    import numpy as np
    assert np.abs(g0 - want_g).max() <= 1e-5 * max(np.abs(want_g).max(), 1e-12)
    assert np.abs(p0 - want_p).max() <= 1e-5
