"""Host-side timeline of a train step (where the host waits, where it runs ahead) -- the evidence behind DESIGN.md section 5.
Usage on a GPU box: python tools/host_timeline.py   (prints medians over 10 drained steps, then a 20-step steady-state run)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import yolox.models.tscd_matching as M
sys.argv = ["bench.py"]
args = bench.parse()
dev = torch.device("cuda", 0)
model = bench.build_model(args.depth, args.width, args.props).to(dev).train()
opt = bench.make_optimizer(model, lr=args.lr); scaler = bench.make_scaler(); ema = bench.FusedEMA(model)
imgs, targets, temb = bench.synth_clip(args.frames, args.size, dev, seed=0)
marks = {}
orig_finish = M.AwarePositionRegMatcher.finish
def finish(self, *a, **k):
    marks["finish_in"] = time.perf_counter()
    r = orig_finish(self, *a, **k)
    marks["finish_out"] = time.perf_counter()
    return r
M.AwarePositionRegMatcher.finish = finish
orig_chain = M.lap_chain_f16
def chain(base_ptr, shapes, kinds, before=None, **kw):
    marks["chain_submit"] = time.perf_counter()
    if before is not None: before()
    marks["copy_done"] = time.perf_counter()
    r = orig_chain(base_ptr, shapes, kinds, None, **kw)
    marks["chain_done"] = time.perf_counter()
    marks["shapes"] = shapes
    return r
M.lap_chain_f16 = chain
def step(rec=None):
    t0 = time.perf_counter()
    out = model(imgs, targets, temb, lframe=args.lframe, gframe=args.frames - args.lframe, resume=False, next_x=imgs)
    t1 = time.perf_counter()
    opt.zero_grad(set_to_none=True); scaler.scale(out["total_loss"]).backward()
    t2 = time.perf_counter()
    scaler.step(opt); scaler.update(); model.join_prefetch(); ema.update()
    t3 = time.perf_counter()
    if rec is not None:
        torch.cuda.synchronize(); t4 = time.perf_counter()
        rec.append((marks["finish_in"] - t0, marks["finish_out"] - t0, t1 - t0, t2 - t0, t3 - t0, t4 - t0))
        rec2.append((marks["chain_submit"] - t0, marks["copy_done"] - t0, marks["chain_done"] - t0))
for _ in range(8): step()
torch.cuda.synchronize()
rec = []; rec2 = []
for _ in range(10): step(rec)
import numpy as np
r = np.median(np.array(rec), 0) * 1e3
print("ms from step start: matcher.finish enter %.2f, leave %.2f | forward returns %.2f | backward returns %.2f | optimizer+EMA queued %.2f | GPU drained %.2f" % tuple(r))
r2 = np.median(np.array(rec2), 0) * 1e3
print("worker: chain submitted %.2f, cost copy landed %.2f, chain solved %.2f ; shapes %s" % (tuple(r2) + (marks["shapes"],)))
# same without the per-step drain (steady state)
torch.cuda.synchronize(); t = time.perf_counter()
waits = []
orig_finish2 = M.AwarePositionRegMatcher.finish
for _ in range(20):
    step(); waits.append(marks["finish_out"] - marks["finish_in"])
torch.cuda.synchronize(); tt = (time.perf_counter() - t) / 20 * 1e3
print("steady ms/step %.2f ; time inside matcher.finish (wait for assignments + queueing CAFM) %.2f ms" % (tt, np.mean(waits) * 1e3))
