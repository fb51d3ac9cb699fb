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


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[2]: DP over 8 ranks, 7 gradient buckets, bf16 messages (bench.py's defaults at N = 8), rehearsed on the
# CPU over gloo with the REAL bucket plan of a depth-12 encoder (14 backward stages -> 7 buckets of two, arena ranges from the
# library's nv_vit_stage_param_range; host-side calls only).  Checked: the buckets tile the arena exactly once; the 8-way sum of
# bf16 messages stays within the stated bound of the fp32 sum; every rank ends with bit-identical values.
DP8_CFG = dict(image_size=32, image_patch_size=8, frames=32, frame_patch_size=8, num_classes=2, dim=96, depth=12, heads=2,
               mlp_dim=384, channels=1, dim_head=64)


def _bucket_plan(n_buckets):
    from neurovit_amd import engine
    cfg = engine.make_config(**DP8_CFG)
    rt = engine.VitRuntime(cfg)
    _, _, total = engine.param_layout(cfg)
    plan = [rt.stage_range(f, l) for f, l in bucket_stages(DP8_CFG["depth"] + 2, n_buckets)]
    return plan, total


def test_bucket_plan_tiles_the_arena_exactly_once():
    for nb in (1, 4, 7, 14):
        plan, total = _bucket_plan(nb)
        cover = torch.zeros(total, dtype=torch.int32)
        for b, e in plan:
            assert 0 <= b < e <= total
            cover[b:e] += 1
        assert int(cover.min()) == 1 and int(cover.max()) == 1, nb
        # backward order: the head's range (arena tail) first, the embedding's (arena head) last
        assert plan[0][1] == total and plan[-1][0] == 0


def _worker8(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan, total = _bucket_plan(7)
        gen = torch.Generator().manual_seed(1234 + rank)
        grads = torch.randn(total, generator=gen) * (10.0 ** torch.randint(-3, 2, (total,), generator=gen).float())   # five decades of magnitude
        exact = grads.double().clone()
        dist.all_reduce(exact)                                    # the sum as float64 messages
        abs_sum = grads.abs().double()
        dist.all_reduce(abs_sum)
        sync = GradSync(None, n_buckets=7, comm_dtype=torch.bfloat16)
        assert sync.world == 8
        sync.begin()
        for b, e in plan:
            sync.bucket_ready(grads, b, e)
        sync.finish()
        assert sync.bytes_reduced == total * 2                    # every element sent once, as bf16
        err = (grads.double() - exact).abs()
        # stated bound: each rank's message is rounded to bf16 (8 significant bits: unit roundoff 2^-8) and each of the 7 additions
        # of the reduction rounds its partial sum to bf16 again: |error| <= 8 * 2^-8 * sum_i |g_i| = 2^-5 * sum_i |g_i| per element
        bound = 2.0 ** -5 * abs_sum + 1e-30
        worst = float((err / bound).max())
        rms = float(err.norm() / exact.norm())
        same = [torch.empty_like(grads) for _ in range(world)]
        dist.all_gather(same, grads)
        identical = all(torch.equal(same[0], t) for t in same)
        q.put((rank, worst, rms, identical))
    finally:
        dist.destroy_process_group()


def test_gradsync_gloo_world8_bf16_messages_7_buckets():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(8))
    for rank, worst, rms, identical in res:
        assert identical, "replicas diverged"
        assert worst <= 1.0, (rank, worst)                        # within the stated elementwise bound
        assert rms < 2.0 ** -7, (rank, rms)                       # and far inside it on average (relative L2 of the whole arena)


# ---------------------------------------------------------------------------------------------------------------------
# the reduce-scatter + all-gather form and the one-hop (all-to-all, local rank-order sum, all-to-all) form of the
# bucket reduction against the plain all-reduce, on gloo.  fp32 messages: every form must leave the SAME values on every rank;
# "one_hop" must equal the left-to-right rank-order sum bit for bit (that is its definition: it is deterministic by construction),
# every form equals the all-reduce bit for bit at world 2 (a + b is commutative) and to fp32 rounding at world 8.  bf16 messages:
# "one_hop" rounds once (sum in fp32) and must be at least as close to the float64 sum as the all-reduce of bf16 messages.
def _worker_algos(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 10007                                                 # not a multiple of the world size or of the 8-element shard alignment
        gen = torch.Generator().manual_seed(77 + rank)
        base = torch.randn(n, generator=gen) * (10.0 ** torch.randint(-2, 2, (n,), generator=gen).float())
        everyone = [torch.empty_like(base) for _ in range(world)]
        dist.all_gather(everyone, base)
        fold = everyone[0].clone()
        for r in range(1, world):
            fold = fold + everyone[r]                             # left-to-right rank-order sum in fp32
        exact = torch.stack(everyone).double().sum(0)
        out = {}
        buckets = [(6000, n), (2500, 6000), (0, 2500)]
        for dtype in (torch.float32, torch.bfloat16):
            for algo in GradSync.ALGOS:
                g = base.clone()
                sync = GradSync(None, n_buckets=3, comm_dtype=dtype, algo=algo)
                sync.begin()
                for b, e in buckets:
                    sync.bucket_ready(g, b, e)
                sync.finish()
                same = [torch.empty_like(g) for _ in range(world)]
                dist.all_gather(same, g)
                out[(str(dtype), algo)] = dict(identical=all(torch.equal(same[0], t) for t in same),
                                               eq_fold=bool(torch.equal(g, fold)),
                                               err=float((g.double() - exact).abs().max() / exact.abs().max()),
                                               rms=float((g.double() - exact).norm() / exact.norm()), g=g)
        f32 = {a: out[(str(torch.float32), a)] for a in GradSync.ALGOS}
        res = dict(rank=rank,
                   identical=all(v["identical"] for v in out.values()),
                   one_hop_is_rank_order_fold=f32["one_hop"]["eq_fold"],
                   f32_bitwise_vs_allreduce={a: bool(torch.equal(f32[a]["g"], f32["allreduce"]["g"])) for a in ("rs_ag", "one_hop")},
                   f32_err={a: f32[a]["err"] for a in GradSync.ALGOS},
                   bf16_rms={a: out[(str(torch.bfloat16), a)]["rms"] for a in GradSync.ALGOS})
        q.put(res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gradsync_reduce_scatter_and_one_hop_forms_against_allreduce(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker_algos, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r["rank"] for r in res) == list(range(world))
    for r in res:
        assert r["identical"], "replicas diverged"
        assert r["one_hop_is_rank_order_fold"], "the one-hop form must be the rank-order fp32 sum, bit for bit"
        for a, err in r["f32_err"].items():
            assert err < 1e-6, (a, err)                           # fp32 messages: every form is the sum to fp32 rounding
        if world == 2:
            assert all(r["f32_bitwise_vs_allreduce"].values()), r["f32_bitwise_vs_allreduce"]
        # bf16 messages: one rounding per rank's message + (one_hop) one of the result, against W - 1 roundings of partial sums
        assert r["bf16_rms"]["one_hop"] <= r["bf16_rms"]["allreduce"] * 1.05 + 1e-9, r["bf16_rms"]
        assert r["bf16_rms"]["rs_ag"] < 2.0 ** -6 and r["bf16_rms"]["one_hop"] < 2.0 ** -7, r["bf16_rms"]


def test_gradsync_rejects_an_unknown_algorithm():
    with pytest.raises(ValueError, match="algo must be one of"):
        GradSync(None, algo="butterfly")
