"""Replica-parallel plumbing for N > 1 (one process per GPU, `torch.distributed`; backend "nccl" is
RCCL on ROCm, "gloo" for the CPU tests).

The decoder hot path shards by SAMPLE exactly like the reference (samples_per_gpu=1 under DDP,
projects/mmdet3d_plugin/apis/mmdet_distill_train.py:62-82): inference has no exchange step inside
the path, so the only collectives are a barrier and a MAX-reduce of the elapsed time.
"""
import os
import time

import torch
import torch.distributed as dist


def env():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def init(backend=None, device=None):
    """Initialise the default process group when WORLD_SIZE > 1.  Returns (rank, world)."""
    rank, _, world = env()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if backend is None:
            backend = 'nccl' if (device is not None and torch.device(device).type == 'cuda') else 'gloo'
        kw = {}
        if backend == 'nccl' and device is not None:
            kw['device_id'] = torch.device(device)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shutdown():
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def sample_indices(num_samples, rank, world):
    """Samples of this rank: r, r+W, r+2W, ... (DistributedSampler order without shuffling)."""
    return list(range(rank, num_samples, world))


def sample_seed(base_seed, rank):
    """Synthetic-input seed of the sample a rank processes (1000+config+rank)."""
    return base_seed + rank


def sync(device=None):
    """barrier + device synchronise on both sides (the bench contract's bracket)."""
    cuda = device is not None and torch.device(device).type == 'cuda'
    if cuda:
        torch.cuda.synchronize(device)
    if dist.is_initialized():
        dist.barrier()
    if cuda:
        torch.cuda.synchronize(device)


def max_over_ranks(value, device=None):
    if not dist.is_initialized():
        return float(value)
    dev = device if (device is not None and dist.get_backend() == 'nccl') else 'cpu'
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def timed_steps(run, steps, warmup, device=None):
    """`warmup` untimed calls of run(), then exactly `steps` timed calls bracketed by sync();
    returns the MAX elapsed seconds over ranks."""
    for _ in range(warmup):
        run()
    sync(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    sync(device)
    return max_over_ranks(time.perf_counter() - t0, device)


def aggregate_throughput(units_per_rank_step, steps, world, elapsed):
    """Whole-job units/s: every rank processed units_per_rank_step * steps units in `elapsed` (max) s."""
    return units_per_rank_step * steps * world / elapsed


class FlatGradAllReducer:
    """Data-parallel gradient averaging with ONE collective per step.

    The reference trains under MMDistributedDataParallel (projects/mmdet3d_plugin/apis/
    mmdet_distill_train.py:78-82): bucketed NCCL all-reduces of ~25 MB plus two scalar all-reduces per
    decoder layer in the loss.  On MI355X the 8 GPUs are fully connected by point-to-point xGMI links
    (7 x ~153 GB/s per GPU), a ring is per-link bound and small collectives are latency bound, so the
    gradients of the whole module are packed into one contiguous fp32 buffer and reduced with a single
    RCCL all-reduce (SUM, then scaled by 1/world) - fewer, larger collectives.  Extra scalars (e.g. the
    loss normalisers the reference reduces one by one) can ride in the same buffer via `extras`.
    """

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None
        self.views = None

    def _buffer(self, like, extra):
        n = self.numel + extra
        if self.flat is None or self.flat.numel() != n or self.flat.device != like.device:
            self.flat = torch.zeros(n, dtype=torch.float32, device=like.device)
            self.views = None
        return self.flat

    def _bind(self, flat):
        """Make every parameter's .grad a view of the flat buffer (keeping its value): autograd then accumulates
        straight into the buffer and a step needs no per-parameter gather / scatter copies (two small launches per
        parameter otherwise - hundreds per step, more than the decoder's own kernels)."""
        if self.views is None:
            self.views, off = [], 0
            for p in self.params:
                n = p.numel()
                self.views.append(flat[off:off + n].view_as(p))
                off += n
        for p, v in zip(self.params, self.views):
            if p.grad is v:
                continue
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
            p.grad = v

    @torch.no_grad()
    def bind(self):
        """Allocate the flat buffer and bind every .grad to it now (parameters without a gradient get zeros).  Needed
        before a backward pass is captured in a hipGraph: the capture records the addresses it accumulates into."""
        self._bind(self._buffer(self.params[0], 0 if self.flat is None else self.flat.numel() - self.numel))

    @torch.no_grad()
    def zero_grad(self):
        """Zero all gradients with one launch (bound buffer); falls back to per-parameter zeroing before the first step."""
        if self.flat is not None and self.views is not None and all(p.grad is v for p, v in zip(self.params, self.views)):
            self.flat.zero_()
            return
        for p in self.params:
            if p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def reduce(self, extras=None):
        """Average .grad of every parameter over all ranks (missing grads count as zero).
        extras: optional 1-D float tensor reduced (summed, NOT averaged) in the same collective; returned."""
        like = self.params[0]
        ne = 0 if extras is None else extras.numel()
        flat = self._buffer(like, ne)
        self._bind(flat)
        if ne:
            flat[self.numel:].copy_(extras.reshape(-1).float())
        world = dist.get_world_size() if dist.is_initialized() else 1
        if world > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat[:self.numel].mul_(1.0 / world)
        return flat[self.numel:].clone() if ne else None

    def bytes_per_step(self):
        return self.numel * 4
