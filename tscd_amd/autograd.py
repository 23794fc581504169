"""Autograd plumbing of the hot path: the eager loss branch."""
import torch


class _EagerBranch(torch.autograd.Function):
    """Runs ``fn(*activations)`` (a scalar loss branch built from modules whose trainable tensors are ``params``) AND its
    backward immediately, inside forward -- so the branch's gradient kernels are queued where the caller is, not where
    autograd's reverse order would put them.  The outer backward only rescales the stored gradients by the upstream
    scalar.  Used by TSCDHead to push the first MCA's backward into the window in which the host solves the matcher's
    assignments and the device would otherwise idle.

    The eager backward is seeded with ``seed``, a 0-dim fp32 DEVICE tensor holding the loss scale the caller expects upstream
    (GradScaler's current scale), so the fp16 gradient kernels see the same magnitudes as in a normal scaled backward and the
    scaler's back-off / growth reaches this branch too; the outer backward multiplies by upstream/seed on the device (no
    sync).  If the branch overflows at the current scale the stored gradients are inf, the outer backward hands inf to the
    scaler, the step is skipped and the scale halves -- and the next forward is seeded with the halved scale.  Parameter
    gradients still reach their AccumulateGrad nodes through the outer graph, so DistributedDataParallel hooks fire as usual."""

    @staticmethod
    def forward(ctx, fn, seed, track, n_act, *tensors):
        acts, params = tensors[:n_act], tensors[n_act:]
        det = [a.detach().requires_grad_(a.requires_grad and a.is_floating_point()) for a in acts]
        with torch.enable_grad():
            loss, extras = fn(*det)
            wrt = [d for d in det if d.requires_grad] + [p for p in params if p.requires_grad]
            seed_now = seed.detach().clone()                      # the caller may update `seed` in place before backward
            grads = torch.autograd.grad(loss * seed_now, wrt, allow_unused=True) if wrt else ()
        ctx.mask = [d.requires_grad for d in det] + [p.requires_grad for p in params]
        ctx.grads = grads
        ctx.seed, ctx.track = seed_now, seed if track else None
        ctx.mark_non_differentiable(*extras)
        return (loss.detach(),) + tuple(extras)

    @staticmethod
    def backward(ctx, g, *unused):
        scale = g.float() / ctx.seed
        if ctx.track is not None:
            ctx.track.copy_(g.float().abs().clamp(min=1.0))       # follow the upstream loss scale (one step behind)
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
        return (None, None, None, None) + tuple(out)


def eager_branch(fn, activations, params, seed, track=False):
    """-> (loss, *extras).  ``fn(*activations) -> (scalar fp32 loss, tuple of detached extra outputs)``.  ``seed``: 0-dim fp32
    device tensor (see _EagerBranch); ``track``: update it in place with the upstream gradient seen in backward."""
    params = [p for p in params]
    return _EagerBranch.apply(fn, seed, bool(track), len(activations), *activations, *params)
