"""N>1 path on CPU: two gloo ranks wrap the product model the way bench.py does (frozen backbone/base head, the 27
never-used parameter tensors of SURVEY F5 frozen so DDP's bucket set is static) and all-reduce gradients.

The HIP kernels cannot run without a GPU, so the forward is replaced by a surrogate that touches exactly the trainable
parameters; what is covered is the distributed plumbing: rendezvous on 127.0.0.1, DDP construction over the mirror's
module tree, gradient averaging, and that the trainable set equals the set of parameters that receive gradients in the
reference (golden ``train_grad_names``) -- the condition for DDP not to hang on unused parameters.
"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ref_names, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = bench.build_model(0.33, 0.125, "500")
    trainable = {k for k, p in model.named_parameters() if p.requires_grad}
    from tscd_amd import ddp as tscd_ddp
    ddp = tscd_ddp.wrap(model)                             # the product's DDP wiring: flat-arena all-reduce (gloo: divide + sum)
    opt = bench.make_optimizer(model)
    ddp.attach(opt)
    ema = bench.FusedEMA(model)
    # surrogate "step": every trainable parameter contributes, scaled by (rank + 1).  Two steps: the first runs on the provisional
    # one-chunk layout and learns the gradient-ready order (broadcast from rank 0), the second on the 4-chunk arena
    firsts, layouts = [], []
    for step in range(2):
        opt.zero_grad(set_to_none=True)
        loss = sum(p.sum() for p in model.parameters() if p.requires_grad) * float(rank + 1)
        loss.backward()
        g = next(p for p in model.parameters() if p.requires_grad).grad
        firsts.append(float(g.flatten()[0]))
        layouts.append(len(ddp.reducer.chunks))
        # every published gradient is a slice of the arena and equals the mean over the ranks
        assert all(p.grad.untyped_storage().data_ptr() == ddp.reducer.arena.untyped_storage().data_ptr()
                   for p in model.parameters() if p.requires_grad)
        assert all(bool((p.grad == 1.5).all()) for p in model.parameters() if p.requires_grad)
        opt.step()
        ema.update()
    assert layouts[1] == 4, layouts
    g0 = firsts[0] if firsts[0] == firsts[1] else float("nan")
    # the ranks' parameters stay identical
    chk = [float(p.double().sum()) for p in model.parameters() if p.requires_grad]
    both = [None] * world
    dist.all_gather_object(both, chk)
    assert all(b == both[0] for b in both)
    q.put((rank, sorted(trainable) == sorted(ref_names), g0, len(opt.param_groups)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gradient_allreduce(golden):
    cfg, sd, vec, meta = golden
    ref_names = ["backbone." + k[len("backbone."):] if k.startswith("backbone.") else k for k in vec["train_grad_names"].tolist()]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ref_names, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same_set, g0, ngroups in res:
        assert same_set, "trainable set differs from the reference's set of parameters with gradients"
        assert abs(g0 - 1.5) < 1e-6          # mean of (1, 2) over the two ranks
        # bn weights / decayed weights / biases / pg3: the reference's substring test "head.cls_convs" in k also matches
        # head.cls_convs2.* (ovis_tscd_large.py:177), so the trainable cls_convs2 weights train at lr * stem_lr_ratio
        assert ngroups == 4


def _run_bench(args, env_extra):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_flag_launches_the_ranks_itself():
    """`python bench.py --gpus N` with no launcher must start N ranks itself (fresh interpreters, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_*), and the one JSON line carries n_gpus = N and the world size the backend reports.  Dry run: gloo on the
    CPU, no model (the GPU form of the same launch is tests/test_gpu_configs.py::test_bench_two_ranks_without_a_launcher)."""
    r, line = _run_bench(["--gpus", "2"], {"TSCD_BENCH_DRYRUN": "1", "TSCD_BENCH_REHEARSAL": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["world_size_reported_by_backend"] == 2 and line["ranks_in_collective"] == 2
    r, line = _run_bench(["--gpus", "3"], {"TSCD_BENCH_DRYRUN": "1", "TSCD_BENCH_REHEARSAL": "1"})
    assert r.returncode == 0 and line["n_gpus"] == 3 and line["ranks_in_collective"] == 3
    # a single rank needs no spawn and no process group
    r, line = _run_bench([], {"TSCD_BENCH_DRYRUN": "1"})
    assert r.returncode == 0 and line["n_gpus"] == 1


def test_bench_refuses_more_ranks_than_devices():
    """Without the rehearsal switch, --gpus N with fewer than N visible devices exits non-zero instead of printing a 1-GPU number
    labelled as N (this container has no GPU: any N > 1 is too many)."""
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a host with fewer than 2 devices")
    r, line = _run_bench(["--gpus", "2"], {"TSCD_BENCH_DRYRUN": "1"})
    assert r.returncode != 0 and line is None and "device(s) visible" in r.stderr
