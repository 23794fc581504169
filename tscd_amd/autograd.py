"""HIP-forward / recompute-backward bridge.

Every proposal-stage op runs its forward in a hand-written HIP kernel.  Until the matching backward kernel exists
(pending rows in DESIGN.md), gradients are obtained by re-evaluating a device-side PyTorch expression of the SAME
math inside ``backward`` and differentiating that.  The PyTorch expression is never used for a forward result.
"""
import torch


class _HipFwdTorchBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hip_fn, torch_fn, *inputs):
        ctx.set_materialize_grads(False)   # unused outputs (e.g. agg_iou's detached cls branch) get no gradient
        ctx.torch_fn = torch_fn
        ctx.is_tensor = [isinstance(t, torch.Tensor) for t in inputs]
        ctx.save_for_backward(*[t for t in inputs if isinstance(t, torch.Tensor)])
        ctx.others = [t for t in inputs if not isinstance(t, torch.Tensor)]
        out = hip_fn(*inputs)
        ctx.single = not isinstance(out, tuple)
        return out

    @staticmethod
    def backward(ctx, *grads):
        saved = list(ctx.saved_tensors)
        others = list(ctx.others)
        ins, leaf = [], []
        for i, is_t in enumerate(ctx.is_tensor):
            if is_t:
                t = saved.pop(0)
                need = ctx.needs_input_grad[2 + i] and t.is_floating_point()
                t = t.detach().requires_grad_(need)
                if need:
                    leaf.append((i, t))
                ins.append(t)
            else:
                ins.append(others.pop(0))
        with torch.enable_grad():
            outs = ctx.torch_fn(*ins)
            outs = (outs,) if ctx.single else outs
            pairs = [(o, g) for o, g in zip(outs, grads) if g is not None and o is not None and o.requires_grad]
            gin = torch.autograd.grad([o for o, _ in pairs], [t for _, t in leaf], [g.to(o.dtype) for o, g in pairs],
                                      allow_unused=True) if pairs and leaf else [None] * len(leaf)
        result = [None, None] + [None] * len(ctx.is_tensor)
        for (i, _), g in zip(leaf, gin):
            result[2 + i] = g
        return tuple(result)


def hip_op(hip_fn, torch_fn, *inputs):
    """Run ``hip_fn(*inputs)``; if autograd needs it, differentiate ``torch_fn(*inputs)`` in backward."""
    if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in inputs):
        return _HipFwdTorchBwd.apply(hip_fn, torch_fn, *inputs)
    return hip_fn(*inputs)


class _EagerBranch(torch.autograd.Function):
    """Runs ``fn(*activations)`` (a scalar loss branch built from modules whose trainable tensors are ``params``) AND its
    backward immediately, inside forward -- so the branch's gradient kernels are queued where the caller is, not where
    autograd's reverse order would put them.  The outer backward only rescales the stored gradients by the upstream
    scalar.  Used by TSCDHead to push the first MCA's backward into the window in which the host solves the matcher's
    assignments and the device would otherwise idle.

    The eager backward is seeded with ``seed`` (the loss scale the caller expects upstream, e.g. GradScaler's 65536) so
    fp16 gradient kernels see the same magnitudes as in a normal scaled backward; the outer backward multiplies by
    upstream/seed on the device (no sync).  Parameter gradients still reach their AccumulateGrad nodes through the outer
    graph, so DistributedDataParallel hooks fire as usual."""

    @staticmethod
    def forward(ctx, fn, seed, n_act, *tensors):
        acts, params = tensors[:n_act], tensors[n_act:]
        det = [a.detach().requires_grad_(a.requires_grad and a.is_floating_point()) for a in acts]
        with torch.enable_grad():
            loss, extras = fn(*det)
            wrt = [d for d in det if d.requires_grad] + [p for p in params if p.requires_grad]
            grads = torch.autograd.grad(loss * seed, wrt, allow_unused=True) if wrt else ()
        ctx.mask = [d.requires_grad for d in det] + [p.requires_grad for p in params]
        ctx.grads = grads
        ctx.seed = seed
        ctx.mark_non_differentiable(*extras)
        return (loss.detach(),) + tuple(extras)

    @staticmethod
    def backward(ctx, g, *unused):
        scale = g.float() / ctx.seed
        live = [gr for gr in ctx.grads if gr is not None]
        # two fused multi-tensor launches (fp32 parameter gradients, fp16 activation gradients) instead of one per tensor
        f32 = [gr for gr in live if gr.dtype == torch.float32]
        f16 = [gr for gr in live if gr.dtype != torch.float32]
        scaled = {}
        if f32:
            scaled.update(zip(map(id, f32), torch._foreach_mul(f32, scale)))
        if f16:
            scaled.update(zip(map(id, f16), torch._foreach_mul(f16, scale.to(f16[0].dtype))))
        it = iter(ctx.grads)
        out = []
        for m in ctx.mask:
            gr = next(it) if m else None
            out.append(None if gr is None else scaled[id(gr)])
        return (None, None, None) + tuple(out)


def eager_branch(fn, activations, params, seed=65536.0):
    """-> (loss, *extras).  ``fn(*activations) -> (scalar fp32 loss, tuple of detached extra outputs)``."""
    params = [p for p in params]
    return _EagerBranch.apply(fn, float(seed), len(activations), *activations, *params)
