"""The 4D model's temporal head on MI355X (csrc/temporal.hip through the C-ABI) against the CPU oracle
(oracle/ref_cpu.py::temporal_head, float64) and against the stock torch modules the reference is made of
(nn.TransformerEncoderLayer(d_model 2, nhead 2) -> mean -> nn.Linear(2, 2): NeuroEncoder.py:60-66, 207-230).

Tolerances: the kernel is fp32 and differs from torch in summation order only; the oracle runs in float64.  Outputs 1e-5 of the
largest value; gradients 2e-4 of the tensor's largest entry plus a floor of 1e-7 (the "soft" parameter set, whose pre-norm
differences sit at the eps scale so that every path carries signal) or 2e-7 of the head's largest gradient (fixture-scale
parameters).  LayerNorm over TWO features maps a row to (+-1, -+1) whenever its two entries differ by more than sqrt(eps): the
gradient through it is eps / (c^2 + eps) ~ 1e-4 of the upstream gradient.  The kernel evaluates that factor in closed form (a
product, csrc/temporal.hip::ln2_bwd), so fixture-scale parameters are held to the same relative gate as the soft ones; the generic
LayerNorm backward - torch's own fp32 kernels included - leaves cancellation noise of 1e-2 ... 8e-2 of a tensor's largest entry
there (round 3 needed a floor of 2e-5 of the head's largest gradient, 100 x this one).
"""
import numpy as np
import pytest
import torch

import weights as W
from conftest import rel_err, report
from oracle import ref_cpu

pytestmark = pytest.mark.gpu
PRE = "temporal_transformer.transformer.layers.0."


@pytest.fixture(scope="module")
def ops():
    from neurovit_amd._cabi import require_gpu
    require_gpu()
    from neurovit_amd import ops as o
    return o


def _params(seed, soft=False):
    sd = W.make_tensors(W.temporal_param_spec(), seed)
    if soft:   # keep both LayerNorm inputs near the eps scale (see module docstring)
        sd[PRE + "self_attn.out_proj.weight"] *= 3e-3
        sd[PRE + "self_attn.out_proj.bias"] *= 3e-2
        sd[PRE + "norm1.weight"] *= 4e-3
        sd[PRE + "norm1.bias"][:] = 0.05
        sd[PRE + "linear2.weight"] *= 2e-2
        sd[PRE + "linear2.bias"] *= 3e-2
    return sd


def _arena(sd):
    return torch.cat([sd[k].reshape(-1) for k, _, _ in W.temporal_param_spec()]).cuda()


def _oracle(sd, x, dout, drop=None):
    leaves = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    xd = x.double().requires_grad_(True)
    out = ref_cpu.temporal_head(leaves, xd, drop)
    names = [k for k, _, _ in W.temporal_param_spec()]
    grads = torch.autograd.grad(out, [leaves[k] for k in names] + [xd], grad_outputs=dout.double())
    return out.detach(), dict(zip(names, grads[:-1])), grads[-1]


def _check(ops, sd, x, dout, drop, tag, soft):
    arena = _arena(sd)
    p, seed = drop if drop else (0.0, 0)
    out = ops.temporal_head_fwd(x.cuda(), arena, 2048, drop_seed=seed, drop_p=p)
    grads = torch.full_like(arena, float("nan"))
    dx = ops.temporal_head_bwd(x.cuda(), arena, 2048, dout.cuda(), grads, accumulate=False, want_dx=True, drop_seed=seed, drop_p=p)
    ref_out, ref_g, ref_dx = _oracle(sd, x, dout, drop)
    e_out = rel_err(out, ref_out)
    assert e_out < 1e-5, (tag, e_out)
    # fixture-scale parameters saturate both LayerNorms (what flows through them is ~eps / (c^2 + eps) = 1e-4 of the upstream gradient);
    # the closed-form two-feature backward carries no cancellation, so the floor is an fp32 epsilon of the head's largest gradient
    gmax = max(v.abs().max().item() for v in ref_g.values())
    floor = 1e-7 if soft else 2e-7 * max(gmax, 1.0)
    off, worst = 0, 0.0
    for k, shape, _ in W.temporal_param_spec():
        n = int(np.prod(shape))
        g = grads[off:off + n].view(shape).cpu().double()
        off += n
        r = ref_g[k]
        err = (g - r).abs().max().item()
        assert err <= 2e-4 * r.abs().max().item() + floor, (tag, k, err, r.abs().max().item())
        worst = max(worst, err / (r.abs().max().item() + 1e-30))
    assert off == arena.numel()
    e_dx = (dx.cpu().double() - ref_dx).abs().max().item()
    assert e_dx <= 2e-4 * ref_dx.abs().max().item() + floor, (tag, e_dx)
    report(f"temporal head {tag}: out rel {e_out:.2e}, worst parameter-gradient rel {worst:.2e}, dx abs {e_dx:.2e} (max {ref_dx.abs().max().item():.2e})")


@pytest.mark.parametrize("B,T", [(1, 20), (2, 5), (4, 20), (3, 64), (2, 1), (5, 7)])
def test_temporal_head_matches_oracle(ops, B, T):
    g = torch.Generator().manual_seed(100 + 7 * B + T)
    for soft in (False, True):
        sd = _params(31 + B, soft)
        x = torch.randn(B, T, 2, generator=g) * (3e-3 if soft else 1.0)
        dout = torch.randn(B, 2, generator=g)
        _check(ops, sd, x, dout, None, f"B{B} T{T} {'soft' if soft else 'fixture-scale'} parameters", soft)


@pytest.mark.parametrize("p", [0.1, 0.5])
def test_temporal_head_dropout_matches_oracle_masks(ops, p):
    """Train mode: the layer's four dropout sites with the kernel's counter-based masks (restated in the oracle); backward
    recomputes the same masks from the seed."""
    g = torch.Generator().manual_seed(5)
    for soft in (False, True):
        sd = _params(41, soft)
        x = torch.randn(3, 20, 2, generator=g) * (3e-3 if soft else 1.0)
        dout = torch.randn(3, 2, generator=g)
        _check(ops, sd, x, dout, (p, 0x1234567 + int(p * 100)), f"dropout {p} {'soft' if soft else 'fixture-scale'}", soft)
    # the masks do something, and a different seed gives a different output
    arena = _arena(_params(41))
    a = ops.temporal_head_fwd(x.cuda(), arena, 2048, drop_seed=1, drop_p=p)
    b = ops.temporal_head_fwd(x.cuda(), arena, 2048, drop_seed=2, drop_p=p)
    c = ops.temporal_head_fwd(x.cuda(), arena, 2048, drop_seed=1, drop_p=p)
    assert not torch.equal(a, b) and torch.equal(a, c)


def test_temporal_head_equals_stock_modules_and_accumulates(ops):
    """Same parameters in the stock modules the reference instantiates (eval mode): forward and autograd gradients."""
    from torch import nn
    sd = _params(51, soft=True)
    layer = nn.TransformerEncoderLayer(d_model=2, nhead=2, batch_first=True)
    enc = nn.TransformerEncoder(layer, num_layers=1)
    proj = nn.Linear(2, 2)
    enc.load_state_dict({k[len("temporal_transformer.transformer."):]: v for k, v in sd.items() if k.startswith(PRE)})
    proj.load_state_dict({"weight": sd["projection_head.projection_head.weight"], "bias": sd["projection_head.projection_head.bias"]})
    enc.cuda().eval(); proj.cuda()
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(2, 20, 2, generator=g) * 3e-3).cuda()
    dout = torch.randn(2, 2, generator=g).cuda()
    ref = proj(enc(x).mean(dim=1))
    ref.backward(dout)
    arena = _arena(sd)
    out = ops.temporal_head_fwd(x, arena, 2048)
    assert rel_err(out, ref) < 1e-5
    grads = torch.zeros_like(arena)
    ops.temporal_head_bwd(x, arena, 2048, dout, grads, accumulate=False)
    once = grads.clone()
    ops.temporal_head_bwd(x, arena, 2048, dout, grads, accumulate=True)
    assert torch.allclose(grads, 2 * once, rtol=1e-6, atol=0)
    stock = torch.cat([q.grad.reshape(-1) for q in list(enc.parameters()) + list(proj.parameters())])
    assert (once - stock).abs().max().item() <= 2e-4 * stock.abs().max().item() + 1e-7
    # bit-reproducible: no atomics anywhere
    again = torch.zeros_like(arena)
    ops.temporal_head_bwd(x, arena, 2048, dout, again, accumulate=False)
    assert torch.equal(again, once)


def test_temporal_head_rejects_unsupported_geometry(ops):
    arena = _arena(_params(1))
    with pytest.raises(RuntimeError, match="timepoints"):
        ops.temporal_head_fwd(torch.zeros(1, 65, 2, device="cuda"), arena, 2048)
