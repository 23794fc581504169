import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import yolox.models.tscd_matching as M
N = 3000
sys.argv = ["bench.py"]
args = bench.parse()
dev = torch.device("cuda", 0)
model = bench.build_model(args.depth, args.width, args.props).to(dev).train()
opt = bench.make_optimizer(model, lr=args.lr)
scaler = bench.make_scaler()
ema = bench.FusedEMA(model, optimizer=opt)
imgs, t0_, temb = bench.synth_clip(args.frames, args.size, dev, seed=0)
targets = bench.targets_on_top_proposals(model, imgs, t0_, temb, args.lframe, args.frames - args.lframe)
one = torch.ones((), device=dev)
t = time.perf_counter()
for i in range(N):
    model.head.loss_scale = scaler.scale(one)
    out = model(imgs, targets, temb, lframe=args.lframe, gframe=args.frames - args.lframe, resume=False, next_x=imgs)
    opt.zero_grad(set_to_none=True)
    scaler.scale(out["total_loss"]).backward()
    model.check_assignments()
    scaler.step(opt)
    scaler.update()
    ema.update()              # reads the pre-prefetch BatchNorm snapshot (no join with the prefetched pass)
    if i % 500 == 499:
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 500 * 1e3
        print(f"steps {i - 499}-{i}: {dt:.2f} ms/step, loss {float(out['total_loss']):.1f}, scale {scaler.get_scale():.0f}, device mem {torch.cuda.memory_allocated() / 2**20:.0f} MiB "
              f"(reserved {torch.cuda.memory_reserved() / 2**20:.0f}), host RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024:.0f} MiB, "
              f"latch generation {M._latch().generation}, timeouts {M._latch().timeouts}", flush=True)
        t = time.perf_counter()
