"""Data-parallel host logic on CPU: gloo, world_size 2 (the N>1 path of SURVEY 8e without GPUs)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from neurovit_amd.parallel import GradSync, broadcast_parameters, bucket_stages


def test_bucket_stages_cover_all_stages():
    for n_stages in (3, 6, 14, 26):
        for nb in (1, 2, 4, 7, 100):
            b = bucket_stages(n_stages, nb)
            assert b[0][0] == 0 and b[-1][1] == n_stages - 1 and len(b) == min(nb, n_stages)
            assert all(x[1] + 1 == y[0] for x, y in zip(b, b[1:]))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)
        n = 1000
        params = torch.randn(n)
        broadcast_parameters(params)
        grads = torch.randn(n)
        expect = grads.clone()
        dist.all_reduce(expect)
        sync = GradSync(None, n_buckets=3)
        assert sync.world == world and abs(sync.grad_scale - 1.0 / world) < 1e-12
        sync.begin()
        for b, e in [(600, 1000), (250, 600), (0, 250)]:          # buckets arrive in backward order (tail first)
            sync.bucket_ready(grads, b, e)
        sync.finish()
        ok = torch.allclose(grads, expect) and sync.bytes_reduced == n * 4
        # bf16 messages (half the xGMI bytes): the sum is exact up to bf16 rounding of each rank's contribution and of the result,
        # and every rank ends with the SAME values
        g16 = torch.randn(n)
        exact = g16.clone()
        dist.all_reduce(exact)
        s16 = GradSync(None, n_buckets=2, comm_dtype=torch.bfloat16)
        s16.begin()
        for b, e in [(500, 1000), (0, 500)]:
            s16.bucket_ready(g16, b, e)
        s16.finish()
        ok = ok and s16.bytes_reduced == n * 2 and float((g16 - exact).abs().max()) <= 2.0 ** -7 * float(exact.abs().max()) and g16.dtype == torch.float32
        both = [torch.empty_like(g16) for _ in range(world)]
        dist.all_gather(both, g16)
        ok = ok and torch.equal(both[0], both[1])
        # replicas stay identical after a (restated) averaged SGD update
        params -= 0.1 * grads * sync.grad_scale
        gathered = [torch.empty_like(params) for _ in range(world)]
        dist.all_gather(gathered, params)
        ok = ok and all(torch.equal(gathered[0], g) for g in gathered)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_gradsync_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
