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
