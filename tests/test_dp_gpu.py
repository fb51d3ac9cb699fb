"""Data-parallel train step on MI355X, world_size 2 over gloo with both ranks on the one GPU of the test box (the driver's
multi-GPU runs use RCCL; the bucket pipeline, stage ranges, scaling and optimizer plumbing under test are the same code).

  * same batch on both ranks + fp32 messages: (g + g) * 0.5 == g exactly, so parameters after two steps must equal the
    single-process run BIT FOR BIT;
  * different batches: equals the single-process step on the concatenated batch up to fp32 summation order;
  * bf16 messages: replicas identical, result within bf16 rounding of the fp32-message run.
"""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import weights as W

pytestmark = pytest.mark.gpu
SIZE = dict(TRAINING_VIT_DIM=128, TRAINING_VIT_DEPTH=2, TRAINING_VIT_HEADS=2, TRAINING_VIT_MLP_DIM=256)


def _model():
    import neurovit_amd.NeuroEncoder as ne
    cfg = W.neuro_config(32, 8, DEVICE="cuda", TRAINING_LEARNING_RATE=1e-3, TRAINING_WEIGHT_DECAY=1e-2, **SIZE)
    model = ne.NeuroEncoder(cfg)
    model.load_state_dict(W.make_tensors(W.vit_param_spec(**W.MICRO), 1, prefix="volume_encoder.vit3d."), strict=True)
    model.train()
    return model


def _data(seed, B=2):
    return W.make_volume((B, 32, 32, 32), seed).cuda(), (torch.arange(B) % 2).cuda()


def _run_steps(model, batches, **kw):
    from neurovit_amd.trainer import TrainStep
    step = TrainStep(model, **kw)
    for x, y in batches:
        step(x, y)
    torch.cuda.synchronize()
    return model.volume_encoder.vit3d.flat_parameters()[0].detach().cpu().clone()


def _worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        model = _model()
        seeds = (7, 8) if mode == "same" else ((7, 8) if rank == 0 else (17, 18))
        comm = torch.bfloat16 if mode == "bf16" else torch.float32
        p = _run_steps(model, [_data(s) for s in seeds], n_buckets=3, grad_comm_dtype=comm)
        q.put((rank, p.numpy()))          # by value: a tensor would travel as a shared-memory handle that dies with this process
    finally:
        dist.destroy_process_group()


def _spawn(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {rank: torch.from_numpy(arr) for rank, arr in (q.get(timeout=240) for _ in procs)}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_dp2_same_batch_equals_single_process_bitwise():
    res = _spawn("same")
    single = _run_steps(_model(), [_data(7), _data(8)])
    assert torch.equal(res[0], res[1])
    assert torch.equal(res[0], single)


def test_dp2_different_batches_equal_concatenated_batch():
    res = _spawn("diff")
    assert torch.equal(res[0], res[1])
    cat = [(torch.cat([_data(a)[0], _data(b)[0]]), torch.cat([_data(a)[1], _data(b)[1]])) for a, b in ((7, 17), (8, 18))]
    single = _run_steps(_model(), cat)
    start = _model().volume_encoder.vit3d.flat_parameters()[0].detach().cpu()
    upd_dp, upd_1 = res[0] - start, single - start
    # AdamW turns 1e-7-level gradient differences (summation order, bf16 operand rounding of different batch shapes) into
    # occasional sign flips of near-zero gradients: compare the updates in relative L2
    assert float((upd_dp - upd_1).norm() / upd_1.norm()) < 0.05


def test_dp2_bf16_messages_close_to_fp32_messages():
    a, b = _spawn("bf16"), _spawn("diff")
    assert torch.equal(a[0], a[1])
    start = _model().volume_encoder.vit3d.flat_parameters()[0].detach().cpu()
    assert float(((a[0] - start) - (b[0] - start)).norm() / (b[0] - start).norm()) < 0.05


def _model4d():
    """4D model: frozen micro encoder + the temporal head (the only trainable parameters; one 10 280-float arena)."""
    import tempfile
    import neurovit_amd.NeuroEncoder as ne
    with tempfile.TemporaryDirectory() as td:
        torch.save(dict(W.make_tensors(W.vit_param_spec(**W.MICRO), 1, prefix="volume_encoder.vit3d.")), os.path.join(td, "c.pth"))
        cfg = W.neuro_config(32, 8, dim=4, DEVICE="cuda", TRAINING_LEARNING_RATE=1e-2, TRAINING_WEIGHT_DECAY=1e-2, GLOBAL_BASE_PATH=td,
                             BEST_MODEL_PATH="c.pth", **SIZE)
        model = ne.NeuroEncoder(cfg)
    model.load_state_dict(W.make_tensors(W.temporal_param_spec(), 3), strict=False)
    lay = model.temporal_transformer.transformer.layers[0]
    lay.dropout.p = lay.dropout1.p = lay.dropout2.p = 0.0          # deterministic across processes
    lay.self_attn.dropout = 0.0
    model.train(); model.volume_encoder.eval()
    return model


def _run_steps4d(model, seeds, **kw):
    from neurovit_amd.trainer import TrainStep
    step = TrainStep(model, **kw)
    for s in seeds:
        step(W.make_volume((2, 32, 32, 32, 4), s).cuda(), torch.tensor([0, 1], device="cuda"))
    torch.cuda.synchronize()
    return model._temporal_head.flat_parameters()[0].detach().cpu().clone()


def _worker4d(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        q.put((rank, _run_steps4d(_model4d(), (7, 8, 9, 10), accumulation_steps=2).numpy()))
    finally:
        dist.destroy_process_group()


def test_dp2_4d_temporal_head_arena_equals_single_process_bitwise():
    """configs[3] under DP: the encoder is frozen, so the only gradient message is the temporal head's arena - ONE all-reduce of
    10 280 floats per optimizer step (trainer.TrainStep), scaled by 1/world before the fused AdamW.  Same batches on both ranks:
    (g + g) * 0.5 == g exactly, so two optimizer steps (two micro-steps each) must leave the head's parameters bit-identical to
    the single-process run."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 150)
    procs = [ctx.Process(target=_worker4d, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {rank: torch.from_numpy(arr) for rank, arr in (q.get(timeout=240) for _ in procs)}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = _run_steps4d(_model4d(), (7, 8, 9, 10), accumulation_steps=2)
    start = _model4d()._temporal_head.flat_parameters()[0].detach().cpu()
    assert torch.equal(res[0], res[1])
    assert torch.equal(res[0], single)
    assert not torch.equal(single, start)


def _msg_worker(q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + (os.getpid() % 200)))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from neurovit_amd.parallel import GradSync
        from neurovit_amd.trainer import TrainStep
        torch.cuda.set_device(0)
        model = _model()
        vit = model.volume_encoder.vit3d
        step = TrainStep(model)
        step.sync = GradSync(None, n_buckets=3, comm_dtype=torch.bfloat16)
        step.sync.world = 2                # force the bucket pipeline; the group has one rank, so the sum is the identity
        step.sync.write_back = False
        step.world = 1
        step(*_data(7))
        torch.cuda.synchronize()
        msg, grads = step.sync.reduced_buffer(), vit.flat_gradients()
        mirrored = vit.mirrored_ranges()
        covered = sum(e - b for b, e in mirrored)
        q.put((bool(torch.equal(msg, grads.to(torch.bfloat16))), covered, grads.numel(), len(mirrored)))
    finally:
        dist.destroy_process_group()


def test_bf16_message_buffer_is_the_rounded_gradient_arena():
    """bf16 messages are assembled without a cast pass over the arena: the Linear weight gradients arrive in the message buffer from
    their GEMMs' epilogues (nv_vit_backward_stages16), the ranges in between through nv_cast_ranges_bf16.  Every element of the buffer
    must equal the rounded fp32 gradient, and the mirrored ranges must be what the module says they are (4 per block + patch embed)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_msg_worker, args=(q,))
    p.start()
    same, covered, total, count = q.get(timeout=240)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert same
    assert count == 4 * SIZE["TRAINING_VIT_DEPTH"] + 1 and covered / total > 0.8


def test_rccl_process_group_runs_the_bucket_pipeline():
    """RCCL under test (every other DP test uses gloo): a real "nccl" process group - one rank, the box has one GPU -
    with the bucket pipeline forced on as at world > 1: 7 buckets all-reduced by RCCL kernels on the side stream during the staged
    backward, fp32 and bf16 messages, broadcast + barrier.  With one rank the sum is the identity, so the loss curve must equal the
    plain single-process run (exactly for fp32 messages).  Runs tools/rccl_rehearsal.py in a child process (the process group must
    not outlive the test, and a HIP-initialised process must not be re-exec'ed: it is spawned, not exec'ed)."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, REHEARSAL_TIMED_STEPS="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_rehearsal.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "rccl rehearsal ok" in r.stdout
    assert "torch.float32" in r.stdout and "torch.bfloat16" in r.stdout


def test_native_dp_step_on_a_one_rank_rccl_communicator_equals_the_single_process_step(monkeypatch):
    """The data-parallel form of the NATIVE step (nv_vit_train_step + nv_dp_plan: backward in stage groups, RCCL all-reduce of every
    gradient bucket issued from native code on its own stream, AdamW per bucket behind it or once at the end) on a communicator of
    ONE rank - all this test box offers; the sum over one rank is the identity, so with fp32 messages the parameters after three steps
    must equal the single-process native step BIT FOR BIT in both update placements.  16-bit messages round the gradients once: close.
    (World sizes 2 and 8 of the Python-driven pipeline are covered over gloo, here and in tests/test_parallel_cpu.py; the native
    path differs from it only in who enqueues the same launches and collectives.)"""
    from neurovit_amd.parallel import NativeComm
    from neurovit_amd.trainer import TrainStep
    batches = [_data(7), _data(8), _data(9)]
    single = _run_steps(_model(), batches, fuse_update=0)
    start = _model().volume_encoder.vit3d.flat_parameters()[0].detach().cpu()
    comm = NativeComm("cuda")
    t = torch.arange(1000, dtype=torch.float32, device="cuda")
    assert torch.equal(comm.all_reduce(t.clone()), t)
    comm.close()
    for per_bucket in ("1", "2", "0"):
        monkeypatch.setenv("NEUROVIT_DP_UPDATE_PER_BUCKET", per_bucket)
        model = _model()
        step = TrainStep(model, n_buckets=3, native_dp=True)
        for x, y in batches:
            step(x, y)
        torch.cuda.synchronize()
        assert step.last_path == "native-dp" and step.last_dp["update_per_bucket"] == int(per_bucket) and step.last_dp["messages"] == "fp32"
        assert torch.equal(model.volume_encoder.vit3d.flat_parameters()[0].detach().cpu(), single), f"update_per_bucket = {per_bucket}"
    model = _model()
    step = TrainStep(model, n_buckets=4, native_dp=True, grad_comm_dtype=torch.bfloat16)
    for x, y in batches:
        step(x, y)
    torch.cuda.synchronize()
    assert step.last_dp["messages"] == "16-bit"
    got = model.volume_encoder.vit3d.flat_parameters()[0].detach().cpu()
    assert float(((got - start) - (single - start)).norm() / (single - start).norm()) < 0.05
    # fp16 operands under the dynamic loss scale: fp32 messages, the overflow check on the reduced gradients, one update at the end
    from neurovit_amd import _cabi
    try:
        ref_model = _model(); ref_model.set_operands("fp16")
        ref = _run_steps(ref_model, batches)
        model = _model(); model.set_operands("fp16")
        step = TrainStep(model, n_buckets=3, native_dp=True)
        for x, y in batches:
            step(x, y)
        torch.cuda.synchronize()
        assert step.scaler is not None and step.last_path == "native-dp" and not step.last_dp["update_per_bucket"]
        assert torch.equal(model.volume_encoder.vit3d.flat_parameters()[0].detach().cpu(), ref)
    finally:
        _cabi.set_operand_format("bf16")
