"""DistributedDataParallel wiring for the clip-sharded train step (reference: yolox/core/tscd_trainer.py:171-176, torch DDP over
the process group of yolox/core/launch.py).

One process per GPU, one clip per rank, gradients averaged over RCCL / xGMI by torch's bucketed reducer, overlapped with the
backward.  Two settings matter on the MI355X step (profiles/r02_ddp_ab.txt):

* gradient_as_bucket_view: parameter gradients ARE slices of the reducer's flat buckets (no copy into the bucket, and FusedSGD
  reads them in place);
* a communication hook.  Without one the reducer pre-divides EVERY parameter's gradient by the world size with its own tiny
  kernel (~150 launches per TSCD-L step); with a hook that duty moves to the hook, once per bucket: RCCL averages in the
  collective itself (ReduceOp.AVG, no extra pass over the 79 MB of fp32 gradients); gloo (CPU rehearsal, tests) has no AVG and
  divides the flat bucket once before its sum."""
import torch
import torch.distributed as dist


def _avg_hook(group, bucket):
    buf = bucket.buffer()
    pg = group if group is not None else dist.group.WORLD
    if dist.get_backend(pg) == "nccl":
        fut = dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=pg, async_op=True).get_future()
    else:
        buf.div_(dist.get_world_size(pg))
        fut = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=pg, async_op=True).get_future()
    return fut.then(lambda f: f.value()[0])


def wrap(model, device_ids=None, process_group=None):
    """model -> DistributedDataParallel(model) with the settings above; broadcast_buffers=False as the reference's trainer
    (tscd_trainer.py:231): each rank keeps the BatchNorm running statistics of its own clips."""
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=device_ids, broadcast_buffers=False,
                                                    gradient_as_bucket_view=True, process_group=process_group)
    ddp.register_comm_hook(process_group, _avg_hook)
    return ddp
