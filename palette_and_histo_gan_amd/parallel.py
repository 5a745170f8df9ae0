"""Data parallelism over the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" is RCCL
over xGMI on ROCm) used purely as the collective transport.

The reference is single-device (no tf.distribute); what a global-batch step must compute under
sharding is derived in SURVEY.md 8e:
  * InstanceNorm, dropout and both networks are per-sample, the losses are element means, so every rank
    evaluates its shard with the GLOBAL element counts in the loss denominators and the global gradient is the
    SUM over ranks (also correct for ragged shards);
  * dgamma/dbeta sum over batch and space and ride in the same flat gradient buffer;
  * the histogram model's Hellinger loss is sqrt(sum over the GLOBAL batch)/B_global (histogram.py:88-89), so one
    extra scalar all-reduce of the local sum of squares sits between the histogram forward and backward.
Collectives per step: all-reduce SUM of the flat generator gradient buffer (117.2 MB f32) in ~6 buckets of >= 16 MB
that are issued asynchronously as soon as the weight-gradient kernels of a bucket have been launched (the flat buffer
is laid out in backward completion order, engine.ParamStore) and so overlap the rest of the backward pass; then one
all-reduce of the small-tensor tail, the discriminator gradient buffer (36.9 KB) and the loss scalars, which share an
allocation with the generator gradients.
"""
import os

import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, device, backend=None):
        self.device = torch.device(device)
        if backend is None:
            backend = "nccl" if self.device.type == "cuda" else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if not dist.is_initialized():
            kw = {}
            if backend == "nccl":
                kw["device_id"] = self.device
            dist.init_process_group(backend=backend, **kw)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self._handles = []

    def allreduce_async(self, t):
        """SUM all-reduce of `t` in place, asynchronously: the collective is ordered after the work already issued on
        the CURRENT stream and runs on the backend's own stream; wait_all() orders later work after it."""
        self._handles.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True))

    def wait_all(self):
        for h in self._handles:
            h.wait()
        self._handles = []

    def allreduce_grads(self, g_grads, d_grads, losses):
        """SUM over ranks, in place.  Loss partials are already scaled with the global counts."""
        dist.all_reduce(g_grads, op=dist.ReduceOp.SUM)
        dist.all_reduce(d_grads, op=dist.ReduceOp.SUM)
        dist.all_reduce(losses, op=dist.ReduceOp.SUM)

    def allreduce_scalar_sum(self, t):
        """The Hellinger coupling: one f32 scalar summed over ranks, in place."""
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def barrier(self):
        dist.barrier()

    def max_scalar(self, x):
        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def destroy(self):
        if dist.is_initialized():
            dist.destroy_process_group()


def init_data_parallel(device, backend=None):
    return DataParallel(device, backend)


def shard_bounds(global_batch, world, rank):
    """Contiguous split of the global batch; earlier ranks take the remainder of a ragged batch."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
