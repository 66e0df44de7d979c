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


def timed_steps(run, steps, warmup, device=None, stats=None):
    """`warmup` untimed calls of run(), then exactly `steps` timed calls bracketed by sync();
    returns the MAX elapsed seconds over ranks.  `stats` (a dict), when given, also receives every rank's view of the
    job: ranks, the slowest and the fastest rank's seconds up to its own device synchronise (before the closing
    barrier) - a self-describing record for multi-GPU runs."""
    for _ in range(warmup):
        run()
    sync(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    if device is not None and torch.device(device).type == 'cuda':
        torch.cuda.synchronize(device)
    own = time.perf_counter() - t0                            # this rank alone
    sync(device)
    total = max_over_ranks(time.perf_counter() - t0, device)
    if stats is not None:
        stats.update(ranks=dist.get_world_size() if dist.is_initialized() else 1,
                     rank_seconds_max=max_over_ranks(own, device), rank_seconds_min=-max_over_ranks(-own, device))
    return total


def aggregate_throughput(units_per_rank_step, steps, world, elapsed):
    """Whole-job units/s: every rank processed units_per_rank_step * steps units in `elapsed` (max) s."""
    return units_per_rank_step * steps * world / elapsed


def preflight(expected_world, device=None, sizes=(22 << 20, 140 << 20, 330 << 20), iters=5, warmup=2, bucket_plan=None):
    """What a first multi-GPU run should establish BEFORE the step is timed, so that a scaling record separates transport from
    compute: (1) the job really has `expected_world` ranks (raises otherwise); (2) every rank's host, device index, device name
    and PCI bus id (two ranks on one device is a launch mistake that still "works"); (3) one all-reduce alone at each of
    `sizes` bytes of fp32 - defaults: the decoder's gradient buffer (22 MB), a ResNet-50 model's (~140 MB) and a VoVNet-99
    model's (~330 MB; SURVEY.md 8e, apis/mmdet_distill_train.py:78-82 is DDP's bucketed all-reduce of these) - barrier-bracketed,
    MAX over ranks, with the bus bandwidth 2 (N - 1) / N x bytes / t a ring moves per link; (4) the bucket plan of the step.
    Returns a dict (identical on every rank); N = 1: {'world': 1}.  gloo (CPU tensors) runs the same code in the tests."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        if int(expected_world) != 1:
            raise RuntimeError(f'preflight: 1 rank(s) in the process group, the job was asked for {expected_world}')
        return {'world': 1}
    import socket
    cuda = device is not None and torch.device(device).type == 'cuda'
    on_dev = cuda and dist.get_backend() == 'nccl'
    me = {'rank': dist.get_rank(), 'host': socket.gethostname(), 'device': str(device) if device is not None else 'cpu',
          'expected_world': int(expected_world)}
    if cuda:
        props = torch.cuda.get_device_properties(device)
        me.update(name=props.name, compute_units=props.multi_processor_count)
        # a hardware identity of the device, if this torch build exposes one (a build without the PCI attributes must not make
        # every rank of a host look like one device): PCI address, else the UUID, else nothing - the check below then falls
        # back to (host, device index)
        if all(hasattr(props, k) for k in ('pci_domain_id', 'pci_bus_id', 'pci_device_id')):
            me['pci'] = '%04x:%02x:%02x' % (int(props.pci_domain_id) & 0xffff, int(props.pci_bus_id) & 0xff, int(props.pci_device_id) & 0xff)
        elif getattr(props, 'uuid', None) is not None:
            me['pci'] = 'uuid:' + str(props.uuid)
    ranks = [None] * world
    dist.all_gather_object(ranks, me)
    # a rank that was asked for another world size (launched with the wrong --gpus) fails the WHOLE job, on every rank, here - not
    # that rank alone while the others wait for it in the first collective
    bad = [r['rank'] for r in ranks if r['expected_world'] != world]
    if bad:
        raise RuntimeError(f'preflight: {world} rank(s) in the process group, rank(s) {bad} were asked for '
                           f'{sorted({r["expected_world"] for r in ranks if r["expected_world"] != world})}')
    seen = {}
    for r in ranks:
        key = (r['host'], r.get('pci', r['device']))
        if key in seen and on_dev:
            raise RuntimeError(f'preflight: ranks {seen[key]} and {r["rank"]} share device {key} - one process per GPU')
        seen[key] = r['rank']
    out = {'world': world, 'backend': dist.get_backend(), 'ranks': ranks, 'bucket_plan': bucket_plan, 'allreduce': []}
    for nbytes in sizes:
        buf = torch.ones(max(1, int(nbytes) // 4), dtype=torch.float32, device=device if on_dev else 'cpu')
        for _ in range(warmup):
            dist.all_reduce(buf)
            buf.fill_(1.0)
            sync(device if cuda else None)            # (every rank has refilled before anybody reduces again)
        t0 = time.perf_counter()
        for _ in range(iters):
            dist.all_reduce(buf)
        if on_dev:
            torch.cuda.synchronize(device)
        sec = max_over_ranks((time.perf_counter() - t0) / iters, device)
        ok = bool(abs(float(buf[0]) - float(world) ** iters) <= 1e-3 * float(world) ** iters)
        sync(device if cuda else None)
        out['allreduce'].append({'bytes': buf.numel() * 4, 'ms': sec * 1e3, 'algbw_GBps': buf.numel() * 4 / sec / 1e9,
                                 'busbw_GBps': 2 * (world - 1) / world * buf.numel() * 4 / sec / 1e9, 'sum_ok': ok})
        del buf
    return out


class FlatGradAllReducer:
    """Data-parallel gradient averaging over ONE flat fp32 buffer.

    The reference trains under MMDistributedDataParallel (projects/mmdet3d_plugin/apis/
    mmdet_distill_train.py:78-82): bucketed NCCL all-reduces of ~25 MB plus two scalar all-reduces per
    decoder layer in the loss.  On MI355X the 8 GPUs are fully connected by point-to-point xGMI links
    (7 x ~153 GB/s per GPU), a ring is per-link bound and small collectives are latency bound, so the
    gradients of the whole module live in one contiguous fp32 buffer (every .grad is a view of it) and are
    reduced by RCCL all-reduces over slices of that buffer (SUM, then scaled by 1/world):

      * reduce()            - one collective for the whole buffer (fewer, larger collectives: right for the
                              22 MB of the decoder alone);
      * buckets + overlap   - `buckets` = groups of parameters in the order their gradients become final
                              during backward (the last decoder layer first).  The buffer is laid out bucket by
                              bucket; install_hooks() starts the asynchronous all-reduce of a bucket as soon as
                              autograd has accumulated its last gradient, so the collective of layer l travels
                              over xGMI while layer l-1 is still in its backward (what the 140-330 MB of the full
                              model need); finish() waits for the collectives and scales.  reduce_buckets() issues
                              the same slices back to back (after a captured backward, where hooks do not run).
                              All three give bit-identical buffers: a slice-wise SUM is the same SUM.

    Extra scalars (e.g. the loss normalisers the reference reduces one by one) ride behind the gradients
    (`extras`, summed, not averaged).  Their capacity is fixed at construction (`max_extras`): the buffer is
    allocated ONCE and never moves afterwards - a hipGraph capture of the backward records its addresses.
    """

    def __init__(self, params, max_extras=0, buckets=None, align=1):
        params = [p for p in params if p.requires_grad]
        if buckets is not None:
            buckets = [[p for p in b if p.requires_grad] for b in buckets]
            buckets = [b for b in buckets if b]
            seen = {id(p) for b in buckets for p in b}
            rest = [p for p in params if id(p) not in seen]
            if len(seen) != sum(len(b) for b in buckets):
                raise ValueError('a parameter appears in more than one bucket')
            if rest:
                buckets.append(rest)                         # whatever the caller did not place goes last
            params = [p for b in buckets for p in b]
        else:
            buckets = [params]
        self.params = params
        # align (elements): every parameter's slice of the flat buffer starts at a multiple of it (4 = 16 bytes: what
        # flatten_params needs - the library's kernels read weights with 16-byte loads); the padding carries zeros
        self.align = max(int(align), 1)
        self.param_numel = sum(p.numel() for p in self.params)
        self.max_extras = int(max_extras)
        self.flat = None
        self.views = None
        # [start, end) of every bucket inside the flat buffer, and every parameter's offset
        self.bucket_ranges, self._offsets, off = [], [], 0
        for b in buckets:
            start = off
            for p in b:
                self._offsets.append(off)
                off += -(-p.numel() // self.align) * self.align
            self.bucket_ranges.append((start, off))
        self.numel = off                                      # (padding included: the extent that is all-reduced)
        self._bucket_of = {}
        for bi, b in enumerate(buckets):
            for p in b:
                self._bucket_of[id(p)] = bi
        self._bucket_sizes = [len(b) for b in buckets]
        self._pending = None
        self._handles = []
        self._hooks = []

    def _buffer(self, like):
        if self.flat is None:
            self.flat = torch.zeros(self.numel + self.max_extras, dtype=torch.float32, device=like.device)
            self.views = None
        elif self.flat.device != like.device:
            raise RuntimeError('FlatGradAllReducer: the parameters moved to another device after the buffer was bound')
        return self.flat

    def _bind(self, flat):
        """Make every parameter's .grad a view of the flat buffer (keeping its value): autograd then accumulates
        straight into the buffer and a step needs no per-parameter gather / scatter copies (two small launches per
        parameter otherwise - hundreds per step, more than the decoder's own kernels)."""
        if self.views is None:
            self.views = [flat[off:off + p.numel()].view_as(p) for p, off in zip(self.params, self._offsets)]
        for p, v in zip(self.params, self.views):
            if p.grad is v:
                continue
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
            p.grad = v

    @torch.no_grad()
    def bind(self, fuse_weight_grads=False):
        """Allocate the flat buffer (gradients + max_extras) and bind every .grad to it now (parameters without a
        gradient get zeros).  Needed before a backward pass is captured in a hipGraph: the capture records the
        addresses it accumulates into; the buffer is never reallocated afterwards.

        fuse_weight_grads: additionally tell this package's autograd Functions (Linear, LayerNorm, value_proj of the
        aggregates) where a parameter's gradient lives (`param._gd4d_main_grad` = its view of the flat buffer): their
        backward kernels then ADD the weight / bias gradient there themselves and hand autograd nothing for the parameter -
        no fresh gradient tensor and no accumulation launch per parameter (~190 of a decoder step's ~290 elementwise adds).
        After backward, `.grad` (the same view) holds exactly what autograd would have accumulated.  Not with
        install_hooks(): a parameter whose gradient bypasses autograd fires no post-accumulate hook."""
        self._bind(self._buffer(self.params[0]))
        if fuse_weight_grads:
            if self._hooks:
                raise RuntimeError('fuse_weight_grads bypasses autograd\'s accumulation: it cannot be combined with install_hooks()')
            for p, v in zip(self.params, self.views):
                p._gd4d_main_grad = v
            self.fused = True

    @torch.no_grad()
    def flatten_params(self):
        """Make every parameter's storage a view of ONE flat fp32 buffer laid out like the gradient buffer (values kept; call it
        before anything captures parameter addresses): an optimizer step is then one elementwise launch over two flat buffers
        (sgd_step) instead of torch.optim.SGD's four multi-tensor launches over 200+ tensors (~110 us of a 5.4-ms training step)."""
        if getattr(self, 'flat_params', None) is not None:
            return self.flat_params
        if self.params[0].is_cuda and torch.cuda.is_current_stream_capturing():
            # re-pointing the parameters at a buffer allocated (and "filled") inside a capture leaves them on memory no kernel has
            # written - and every graph captured earlier on their old addresses
            raise RuntimeError('FlatGradAllReducer.flatten_params: first call inside a hipGraph capture - call it (or adamw_state() / one '
                               'eager optimizer step) before capturing')
        if self.align % 4:
            raise RuntimeError('FlatGradAllReducer.flatten_params needs align=4 (or a multiple): the kernels read weights with 16-byte loads')
        like = self.params[0]
        fp = torch.zeros(self.numel, dtype=torch.float32, device=like.device)
        for p, off in zip(self.params, self._offsets):
            if p.dtype != torch.float32 or p.device != like.device:
                raise RuntimeError('FlatGradAllReducer.flatten_params: fp32 parameters on one device')
            v = fp[off:off + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
        self.flat_params = fp
        return fp

    @torch.no_grad()
    def sgd_step(self, lr):
        """p -= lr * p.grad for every parameter, one launch (plain SGD: no momentum, no weight decay - what
        torch.optim.SGD(params, lr) computes, bit for bit).  Needs bind() (the gradients in the flat buffer) - and
        flatten_params(), which it calls.  Outside a graph capture the parameters' version counters are advanced as an in-place
        update would (the weight-image caches of the row chains are keyed to them)."""
        fp = self.flatten_params()
        flat = self._buffer(self.params[0])
        if self.views is None or any(p.grad is not v for p, v in zip(self.params, self.views)):
            self._bind(flat)
        fp.add_(flat[:self.numel], alpha=-float(lr))
        if not (fp.is_cuda and torch.cuda.is_current_stream_capturing()):
            bump = getattr(torch.autograd.graph, 'increment_version', None)
            for p in self.params:
                if bump is not None:
                    bump(p)
                else:
                    p.add_(0)

    @torch.no_grad()
    def adamw_step(self, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_norm=35.0):
        """The reference's optimizer step (projects/configs/detr4d/detr4d_res50_deform_pe_testaug_320_fullset_ceph.py:205-213: AdamW
        lr 2e-4, weight decay 0.01, grad_clip max_norm 35 / L2) over the flat parameter and gradient buffers: gd4d_adamw_flat, two
        launches instead of clip_grad_norm_ + torch.optim.AdamW's dozen multi-tensor launches over 230 tensors; the step counter
        lives on the device, so the update can sit inside a replayed hipGraph.  One learning rate for every parameter (the
        decoder's; the reference scales the backbone's by 0.1 - not part of this buffer).  Needs bind(); calls flatten_params().
        After the step self.last_grad_norm (a device scalar) holds the norm before clipping."""
        from . import _lib
        if getattr(self, '_adam', None) is None and self.params[0].is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError('FlatGradAllReducer.adamw_step: first call inside a hipGraph capture (its zero-filled state would be reset by '
                               'every replay) - call adamw_state() (or one eager step) before capturing')
        fp = self.flatten_params()
        flat = self._buffer(self.params[0])
        if self.views is None or any(p.grad is not v for p, v in zip(self.params, self.views)):
            self._bind(flat)
        if getattr(self, '_adam', None) is None:
            if torch.cuda.is_current_stream_capturing():
                # zero-fills recorded into a capture would reset m, v and the device step counter on EVERY replay (each step would
                # silently be Adam's first): the state must exist before the capture starts
                raise RuntimeError('FlatGradAllReducer.adamw_step: first call inside a hipGraph capture - call adamw_state() (or one '
                                   'eager step) before capturing')
            self.adamw_state()
        m, v, state, ws = self._adam
        code = _lib.load().gd4d_adamw_flat(fp.data_ptr(), flat.data_ptr(), m.data_ptr(), v.data_ptr(), state.data_ptr(), ws.data_ptr(),
                                           ws.numel(), self.numel, float(lr), float(betas[0]), float(betas[1]), float(eps),
                                           float(weight_decay), float(max_norm or 0.), torch.cuda.current_stream(fp.device).cuda_stream)
        _lib.check(code, 'gd4d_adamw_flat')
        self.last_grad_norm = state[1]
        if not torch.cuda.is_current_stream_capturing():
            bump = getattr(torch.autograd.graph, 'increment_version', None)
            for p in self.params:
                if bump is not None:
                    bump(p)
                else:
                    p.add_(0)

    @torch.no_grad()
    def adamw_state(self):
        """Allocate AdamW's state (exp_avg, exp_avg_sq, {step, norm} on the device, workspace) - idempotent; call it before a capture
        whose first adamw_step would otherwise be inside it."""
        if getattr(self, '_adam', None) is None:
            from . import _lib
            fp = self.flatten_params()
            self._adam = (torch.zeros_like(fp), torch.zeros_like(fp), torch.zeros(2, device=fp.device, dtype=torch.float32),
                          torch.empty(int(_lib.load().gd4d_adamw_flat_workspace_bytes()), device=fp.device, dtype=torch.uint8))
        return self._adam

    def after_replays(self):
        """A replayed hipGraph updates the parameters without bumping their version counters (the bump is host work, skipped while
        capturing): the owner of the graph calls this after replays and before anything that caches by version reads the parameters
        (weight images of the row chains, the head's position-embedding caches) - versions bumped, image caches forgotten."""
        from . import ops
        bump = getattr(torch.autograd.graph, 'increment_version', None)
        with torch.no_grad():
            for p in self.params:
                if bump is not None:
                    bump(p)
                else:
                    p.add_(0)
        ops.invalidate_chain_images()

    def unfuse(self):
        for p in self.params:
            if hasattr(p, '_gd4d_main_grad'):
                del p._gd4d_main_grad
        self.fused = False

    @torch.no_grad()
    def zero_grad(self):
        """Zero all gradients with one launch (bound buffer); falls back to per-parameter zeroing before the first step."""
        if self.flat is not None and self.views is not None and all(p.grad is v for p, v in zip(self.params, self.views)):
            self.flat.zero_()
            return
        for p in self.params:
            if p.grad is not None:
                p.grad.zero_()

    def _world(self):
        return dist.get_world_size() if dist.is_initialized() else 1

    def _stage_extras(self, flat, extras):
        ne = 0 if extras is None else extras.numel()
        if ne > self.max_extras:
            raise ValueError(f'FlatGradAllReducer: {ne} extras but the buffer was sized for max_extras='
                             f'{self.max_extras}; the bound buffer never grows (its address may be captured in a '
                             f'hipGraph) - pass max_extras at construction')
        if self.max_extras:
            flat[self.numel:].zero_()
        if ne:
            flat[self.numel:self.numel + ne].copy_(extras.reshape(-1).float())
        return ne

    @torch.no_grad()
    def reduce(self, extras=None):
        """Average .grad of every parameter over all ranks (missing grads count as zero) with ONE collective.
        extras: optional 1-D float tensor (<= max_extras elements) reduced (summed, NOT averaged) in the same
        collective; returned."""
        flat = self._buffer(self.params[0])
        self._bind(flat)
        ne = self._stage_extras(flat, extras)
        world = self._world()
        if world > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat[:self.numel].mul_(1.0 / world)
        return flat[self.numel:self.numel + ne].clone() if ne else None

    # ---- bucketed form: the all-reduce of a bucket overlaps the backward of the layers below it ----
    def _launch_bucket(self, bi):
        lo, hi = self.bucket_ranges[bi]
        if self._world() > 1 and hi > lo:
            self._handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        self._launched.append(bi)

    def install_hooks(self):
        """Register post-accumulate hooks: when the last gradient of a bucket has been accumulated during backward, the
        bucket's all-reduce starts (asynchronously, on the backend's own stream).  Call begin_step() before each
        backward and finish() after it."""
        if self._hooks:
            return
        if getattr(self, 'fused', False):
            raise RuntimeError('install_hooks() needs autograd to accumulate the gradients: unfuse() first')
        self.bind()

        def make(p):
            def hook(_):
                if self._pending is None:
                    return
                bi = self._bucket_of[id(p)]
                self._pending[bi] -= 1
                if self._pending[bi] == 0:
                    self._launch_bucket(bi)
            return hook
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(make(p)))

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def begin_step(self):
        """Arm the hooks for one backward pass (zero_grad() first)."""
        self._pending = list(self._bucket_sizes)
        self._handles, self._launched = [], []

    @torch.no_grad()
    def finish(self, extras=None):
        """After backward: launch the buckets whose hooks never completed (parameters without a gradient this step),
        reduce the extras, wait for every collective and scale the gradients by 1/world."""
        flat = self._buffer(self.params[0])
        if self._pending is None:
            self.begin_step()
        for bi in range(len(self.bucket_ranges)):
            if bi not in self._launched:
                self._launch_bucket(bi)
        ne = self._stage_extras(flat, extras)
        world = self._world()
        if world > 1 and self.max_extras:
            self._handles.append(dist.all_reduce(flat[self.numel:], op=dist.ReduceOp.SUM, async_op=True))
        for h in self._handles:
            h.wait()
        self._handles, self._pending = [], None
        if world > 1:
            flat[:self.numel].mul_(1.0 / world)
        return flat[self.numel:self.numel + ne].clone() if ne else None

    @torch.no_grad()
    def reduce_buckets(self, extras=None):
        """The bucketed collectives issued back to back (no hooks: e.g. after a hipGraph replay of the backward)."""
        self._bind(self._buffer(self.params[0]))
        self.begin_step()
        return self.finish(extras)

    def bytes_per_step(self):
        return self.numel * 4

    def describe(self):
        """Self-description for the bench line: bytes per step and per bucket."""
        return dict(allreduce_bytes=self.bytes_per_step() + 4 * self.max_extras,
                    buckets=[(hi - lo) * 4 for lo, hi in self.bucket_ranges])


def decoder_buckets(transformer, *others):
    """Gradient buckets of a Detr3DTransformer in the order backward finalises them: `others` (head branches - their
    gradients arrive first), then the decoder layers from the last to the first, then the rest."""
    buckets = [list(m.parameters()) for m in others if m is not None]
    layers = list(transformer.decoder.layers)
    for layer in reversed(layers):
        buckets.append(list(layer.parameters()))
    return buckets
