#!/usr/bin/env python3
"""Which threads of the training process burn host CPU next to the assignment worker: runs the step of tools/step_timeline.py (plain or, with
TSCD_TIMELINE_DDP=1, under a 1-rank RCCL group) for N steps and prints every thread's user + system time and the core it last ran on."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
sys.argv = ["bench.py"]
args = bench.parse()
dev = torch.device("cuda", 0)
model = bench.build_model(args.depth, args.width, args.props).to(dev).train()
opt = bench.make_optimizer(model, lr=args.lr)
net = model
if os.environ.get("TSCD_TIMELINE_DDP") == "1":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29543")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from tscd_amd import ddp as tscd_ddp
    net = tscd_ddp.wrap(model, device_ids=[0])
    net.attach(opt)
scaler = bench.make_scaler()
ema = bench.FusedEMA(model, optimizer=opt)
imgs, t0_, temb = bench.synth_clip(args.frames, args.size, dev, seed=0)
targets = bench.targets_on_top_proposals(model, imgs, t0_, temb, args.lframe, args.frames - args.lframe)
one = torch.ones((), device=dev)


def step():
    model.head.loss_scale = scaler.scale(one)
    out = net(imgs, targets, temb, lframe=args.lframe, gframe=args.frames - args.lframe, resume=False, next_x=imgs)
    opt.zero_grad(set_to_none=True)
    scaler.scale(out["total_loss"]).backward()
    model.check_assignments()
    scaler.step(opt)
    scaler.update()
    ema.update()


def threads():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            with open(f"/proc/self/task/{tid}/stat") as fh:
                f = fh.read().rsplit(")", 1)[1].split()
            with open(f"/proc/self/task/{tid}/comm") as fh:
                name = fh.read().strip()
            out[int(tid)] = (name, (int(f[11]) + int(f[12])) / os.sysconf("SC_CLK_TCK"), int(f[36]))
        except OSError:
            pass
    return out


for _ in range(80):
    step()
torch.cuda.synchronize()
a = threads()
t0 = time.perf_counter()
for _ in range(N):
    step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
b = threads()
print(f"{N} steps: {dt / N * 1e3:.2f} ms per step; affinity: {len(os.sched_getaffinity(0))} cores")
rows = sorted(((b[t][1] - a.get(t, (None, 0, 0))[1], b[t][0], t, b[t][2]) for t in b), reverse=True)
for cpu, name, tid, core in rows[:14]:
    print(f"  {name:18s} tid {tid:7d}  {cpu / dt * 100:6.1f} % of a core   last on core {core}")
