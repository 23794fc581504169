"""Deferred, batched weight gradients of the proposal stage's linears (tscd_amd/ops.py: _WgradBatch; conv_wgrad.hip:
conv_wgrad_glds_jobs_kernel) and the one-launch gradient check / gather (elementwise.hip: grad_gather_check_kernel).

Reference semantics: autograd of nn.Linear (dW = dY^T X, fp32 accumulation) for the linears of tscd_head.py:123,130-133,
post_trans.py:566-577, tscd_matching.py:31-60; GradScaler's inf check (tscd_trainer.py:171-176).  Tolerance: fp16 operands, fp32
accumulation over <= 8000 rows -> 2e-3 of max|ref| (the immediate split-K path sums in another order, so the two HIP paths are compared
with the fp32 torch product, not bit for bit)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(2000, 1024, 1024), (2000, 256, 256), (8000, 256, 512), (2000, 768, 1024), (2000, 1024, 25), (2000, 1024, 1), (2000, 1024, 4),
          (37, 256, 256), (500, 512, 512), (2000, 256, 1024)]


def _layers(seed=0):
    g = torch.Generator().manual_seed(seed)
    out = []
    for n, i, o in SHAPES:
        lin = torch.nn.Linear(i, o).cuda()
        x = (torch.randn(n, i, generator=g) * 0.5).half().cuda()
        gy = (torch.randn(n, o, generator=g) * 0.1).half().cuda()
        out.append((lin, x, gy))
    return out


def _run(layers, defer, monkeypatch, twice=False):
    from tscd_amd import ops
    from yolox.models.tscd_matching import linear
    monkeypatch.setattr(ops, "DEFER_LINEAR_WGRAD", defer)
    for lin, _, _ in layers:
        lin.weight.grad = lin.bias.grad = None
    xs = [x.clone().requires_grad_() for _, x, _ in layers]
    loss = 0
    for (lin, _, gy), x in zip(layers, xs):
        y = linear(x, lin)
        loss = loss + (y.float() * gy.float()).sum()
        if twice:                      # the same parameter contributes twice in one pass: autograd adds the contributions
            loss = loss + (linear(x * 0.5, lin).float() * gy.float()).sum()
    loss.backward()
    torch.cuda.synchronize()
    return [(lin.weight.grad.clone(), lin.bias.grad.clone(), x.grad.clone()) for (lin, _, _), x in zip(layers, xs)]


@pytest.mark.parametrize("twice", [False, True])
def test_deferred_linear_weight_gradients_match_fp32_products(monkeypatch, twice):
    from tscd_amd import ops
    layers = _layers()
    launched = []
    real = ops.call

    def spy(name, *a):
        launched.append(name)
        return real(name, *a)

    monkeypatch.setattr(ops, "call", spy)
    got = _run(layers, True, monkeypatch, twice)
    assert launched.count("tscd_wgrad_batch_f16") >= 1
    if not twice:
        assert launched.count("tscd_wgrad_batch_f16") == 1          # all ten weight gradients in ONE launch
    imm = _run(layers, False, monkeypatch, twice)
    k = 1.5 if twice else 1.0
    for (lin, x, gy), (dw, db, dx), (dw_i, db_i, dx_i) in zip(layers, got, imm):
        ref_w = gy.float().t() @ x.float() * (1.0 + (0.5 if twice else 0.0))
        ref_b = gy.float().sum(0) * (2.0 if twice else 1.0)
        sw, sb = float(ref_w.abs().max()) + 1e-12, float(ref_b.abs().max()) + 1e-12
        assert float((dw - ref_w).abs().max()) <= 2e-3 * sw + 1e-5, (tuple(x.shape), float((dw - ref_w).abs().max()), sw)
        assert float((dw_i - ref_w).abs().max()) <= 2e-3 * sw + 1e-5
        assert float((db - ref_b).abs().max()) <= 2e-3 * sb + 1e-5
        assert float((db_i - ref_b).abs().max()) <= 2e-3 * sb + 1e-5        # (immediate: two-stage column sums; deferred: the batched launch)
        assert torch.equal(dx, dx_i)                                         # data gradients take the same path either way
        assert lin.weight.grad.shape == lin.weight.shape and lin.weight.grad.is_contiguous()


def test_deferred_gradient_is_adopted_not_cloned(monkeypatch):
    """AccumulateGrad must adopt the registered tensor (no copy launch, and the flush writes where .grad points)."""
    from tscd_amd import ops
    layers = _layers(1)[:3]
    seen = []
    real_compute = ops._WgradBatch.compute

    def compute(self):
        seen.extend(dw for prm, dw in self.check if prm.dim() == 2)            # the weights' entries (the biases are registered too)
        return real_compute(self)

    monkeypatch.setattr(ops._WgradBatch, "compute", compute)
    _run(layers, True, monkeypatch)
    assert len(seen) == 3
    registered = {dw.untyped_storage().data_ptr() for dw in seen}          # (registered in backward order)
    for lin, _, _ in layers:
        assert lin.weight.grad.untyped_storage().data_ptr() in registered


def test_grad_check_and_gather_one_launch():
    """tscd_grad_gather_check_f32 (FusedSGD.check_finite): found_inf stays 0 on finite gradients, becomes 1 for one Inf / NaN anywhere;
    the gather form copies (scaled) into an arena."""
    import numpy as np
    from tscd_amd import ops
    from tscd_amd._lib import lib
    g = torch.Generator().manual_seed(0)
    sizes = [5, 1024, 70001, 3, 262144 + 17]
    grads = [torch.randn(n, generator=g).cuda() for n in sizes]
    ch = lib.tscd_ema_chunk_elems()
    arena = torch.zeros(sum(sizes) + 8, device="cuda")

    def table(with_dst):
        rows, off = [], 0
        for pi, t in enumerate(grads):
            for o in range(0, t.numel(), ch):
                rows.append(((arena.data_ptr() + 4 * (off + o)) if with_dst else 0, 4 * o, min(ch, t.numel() - o), pi))
            off += t.numel()
        return torch.from_numpy(np.array(rows, dtype=np.int64)).cuda(), len(rows)

    base = torch.tensor([t.data_ptr() for t in grads], dtype=torch.int64).cuda()
    fi = torch.zeros((), device="cuda")
    tab, n = table(False)
    ops.call("tscd_grad_gather_check_f32", tab.data_ptr(), n, base.data_ptr(), 1.0, fi.data_ptr(), ops._stream())
    assert float(fi) == 0.0
    tab2, n2 = table(True)
    ops.call("tscd_grad_gather_check_f32", tab2.data_ptr(), n2, base.data_ptr(), 0.5, 0, ops._stream())
    assert torch.equal(arena[:sum(sizes)], torch.cat(grads) * 0.5)
    for bad, where in ((float("inf"), (2, 69999)), (float("nan"), (4, 262144 + 16)), (-float("inf"), (0, 0))):
        grads[where[0]][where[1]] = bad
        fi.zero_()
        ops.call("tscd_grad_gather_check_f32", tab.data_ptr(), n, base.data_ptr(), 1.0, fi.data_ptr(), ops._stream())
        assert float(fi) == 1.0, (bad, where)
        grads[where[0]][where[1]] = 0.0


def test_fused_grad_scaler_skips_and_backs_off_like_torch():
    """tscd_amd.optim.GradScaler + FusedSGD against torch.amp.GradScaler + torch SGD: same parameters after a clean step, a skipped
    step (Inf gradient) and the scale after its back-off."""
    from tscd_amd.optim import FusedSGD, GradScaler
    g = torch.Generator().manual_seed(3)
    w0 = [torch.randn(300, 70, generator=g), torch.randn(70, generator=g)]
    gr = [[torch.randn(300, 70, generator=g), torch.randn(70, generator=g)] for _ in range(3)]
    gr[1][0][5, 5] = float("inf")

    def run(opt_cls, scaler):
        ps = [torch.nn.Parameter(w.clone().cuda()) for w in w0]
        opt = opt_cls(ps, lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-3)
        scales = []
        one = torch.ones((), device="cuda")
        for step in range(3):
            sc = float(scaler.scale(one))                 # (also initialises the scaler's device state, as scale(loss) does in a step)
            for p, q in zip(ps, gr[step]):
                p.grad = (q * sc).cuda()
            scaler.step(opt)
            scaler.update()
            scales.append(scaler.get_scale())
        return [p.detach().clone() for p in ps], scales

    a, sa = run(FusedSGD, GradScaler("cuda", init_scale=1024.0))
    b, sb = run(torch.optim.SGD, torch.amp.GradScaler("cuda", init_scale=1024.0))
    assert sa == sb == [1024.0, 512.0, 512.0]
    for x, y in zip(a, b):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shape", [(2, 18, 18, 64, 64, 3, 1), (4, 36, 36, 256, 256, 3, 2), (3, 20, 24, 96, 32, 1, 1), (2, 9, 9, 768, 768, 1, 1),
                                   (16, 72, 72, 64, 128, 3, 1)])
def test_deferred_conv_weight_gradients_match_the_immediate_kernels_and_torch(monkeypatch, shape):
    """ops.conv_wgrad(..., grad_params=...) inside a backward pass (registered, computed by the pass's batched launch) against the
    immediate path and torch's conv weight gradient in fp32: 3x3 and 1x1, grouped (the towers' second convs), split and unsplit."""
    import torch.nn.functional as F
    from tscd_amd import ops
    B, H, W, cin, cout, k, groups = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(B, H, W, groups * cin, generator=g) * 0.5).half().cuda()
    dy = (torch.randn(B, H, W, groups * cout, generator=g) * 0.1).half().cuda()
    params = [torch.nn.Parameter(torch.zeros(cout, cin, k, k, device="cuda")) for _ in range(groups)]

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, *ps):
            return torch.zeros((), device="cuda")

        @staticmethod
        def backward(ctx, g_):
            dw = ops.conv_wgrad(x, dy, cin, cout, k, groups=groups, grad_params=list(params))
            return tuple(dw[i * cout:(i + 1) * cout] for i in range(groups))

    launched = []
    real = ops.call
    monkeypatch.setattr(ops, "call", lambda name, *a: (launched.append(name), real(name, *a))[1])
    Fn.apply(*params).backward()
    torch.cuda.synchronize()
    assert launched.count("tscd_wgrad_batch_f16") == 1 and not any(n.startswith("tscd_conv_wgrad") for n in launched)
    imm = ops.conv_wgrad(x, dy, cin, cout, k, groups=groups)
    torch.cuda.synchronize()
    for i, p in enumerate(params):
        xi = x[..., i * cin:(i + 1) * cin].permute(0, 3, 1, 2).float()
        dyi = dy[..., i * cout:(i + 1) * cout].permute(0, 3, 1, 2).float()
        ref = torch.nn.grad.conv2d_weight(xi, (cout, cin, k, k), dyi, padding=(k - 1) // 2)
        sc = float(ref.abs().max()) + 1e-12
        assert float((p.grad - ref).abs().max()) <= 2e-3 * sc + 1e-5, (shape, i, float((p.grad - ref).abs().max()), sc)
        assert float((imm[i * cout:(i + 1) * cout] - ref).abs().max()) <= 2e-3 * sc + 1e-5
