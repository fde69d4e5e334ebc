"""Image-parallel data parallelism: one process per GPU, RCCL (torch.distributed backend 'nccl' on ROCm) gradient
all-reduce only -- the hot-path kernels are per image and need no collective.

Mirrors mmdet/core/utils/dist_utils.py:9-56 (allreduce_grads / _allreduce_coalesced / DistOptimizerHook),
mmdet/apis/train.py:35-82 (parse_losses, batch_processor) and the rank-sharding rule of
mmdet/datasets/loader/sampler.py:78-164 (each rank takes an equal, padded slice of a seeded permutation).
"""
from collections import OrderedDict

import torch
import torch.distributed as dist
from torch._utils import _flatten_dense_tensors, _take_tensors, _unflatten_dense_tensors
from torch.nn.utils import clip_grad


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_dist(backend='nccl', **kwargs):
    """torch.distributed.run style launch (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment)."""
    import os
    rank = int(os.environ['RANK'])
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', rank % max(torch.cuda.device_count(), 1))))
    dist.init_process_group(backend=backend, **kwargs)


def _allreduce_coalesced(tensors, world_size, bucket_size_mb=-1):
    if bucket_size_mb > 0:
        bucket_size_bytes = bucket_size_mb * 1024 * 1024
        buckets = _take_tensors(tensors, bucket_size_bytes)
    else:
        buckets = OrderedDict()
        for tensor in tensors:
            tp = tensor.type()
            if tp not in buckets:
                buckets[tp] = []
            buckets[tp].append(tensor)
        buckets = buckets.values()
    for bucket in buckets:
        flat_tensors = _flatten_dense_tensors(bucket)
        dist.all_reduce(flat_tensors)
        flat_tensors.div_(world_size)
        for tensor, synced in zip(bucket, _unflatten_dense_tensors(flat_tensors, bucket)):
            tensor.copy_(synced)


def allreduce_grads(params, coalesce=True, bucket_size_mb=-1):
    """Average the gradients over the ranks.  On MI355X one flat all-reduce per dtype keeps RCCL on large messages:
    xGMI is point-to-point (7 links x ~153 GB/s per GPU), ~146 MB of fp32 gradients for R-50 is ~1-2 ms."""
    grads = [param.grad.data for param in params if param.requires_grad and param.grad is not None]
    world_size = dist.get_world_size()
    if coalesce:
        _allreduce_coalesced(grads, world_size, bucket_size_mb)
    else:
        for tensor in grads:
            dist.all_reduce(tensor.div_(world_size))


class OverlappedGradientReducer(object):
    """Bucketed gradient averaging that runs WHILE backward is still producing gradients (the role of the DDP wrapper in
    mmdet/apis/train.py:137-141), sized for MI355X: xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring
    all-reduce is per-link bound and wants few, large messages -- 32 MB buckets: R-50's ~146 MB of fp32 gradients are
    5 collectives, each long enough to stay bandwidth-bound, the first issued after the head's backward.

    * every bucket is ONE flat buffer; `param.grad` are views into it, so nothing is copied before or after a collective;
    * parameters are bucketed in reverse registration order (~ the order backward produces them); a per-parameter
      post-accumulate hook counts arrivals, a complete bucket is pre-divided by the world size and all-reduced with
      async_op=True (RCCL's own stream: it overlaps the rest of backward);
    * buckets are ALWAYS issued in index order and `finish()` issues whatever is left (parameters that took no part in
      this iteration on this rank contribute zeros), so every rank issues the same sequence of collectives even when
      their autograd graphs differ (an image without positives);
    * every bucket carries one "took part" flag per parameter behind its gradients, reduced by the same collective: a
      parameter no rank produced a gradient for is found without an extra collective (one small host read in finish())."""

    def __init__(self, params, bucket_cap_mb=32):
        self.params = [p for p in params if p.requires_grad]
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        cap = int(bucket_cap_mb * 1024 * 1024)
        self.buckets = []                                 # dicts: flat, params, pending, launched, handle
        cur, cur_bytes, key = [], 0, None
        for p in reversed(self.params):
            k = (p.dtype, p.device)
            nbytes = p.numel() * p.element_size()
            if cur and (k != key or cur_bytes + nbytes > cap):
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
            key = k
        if cur:
            self._close(cur)
        self._bucket_of, self._offset = {}, {}
        for bi, b in enumerate(self.buckets):
            off = 0
            for k, p in enumerate(b['params']):
                self._bucket_of[id(p)] = bi
                self._offset[id(p)] = (off, k)                 # element offset of the gradient view, flag slot
                off += p.numel()
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._next = 0
        self.zero_grad()

    def _close(self, plist):
        n = sum(p.numel() for p in plist)
        flat = torch.zeros(n + len(plist), dtype=plist[0].dtype, device=plist[0].device)     # gradients | one flag per parameter
        # host staging for the flag region: PINNED for a GPU bucket, so the copy issued from inside the autograd hook is a
        # real asynchronous H2D (a pageable source makes non_blocking a no-op and blocks the autograd thread until the
        # bucket's preceding backward kernels have run -- one host sync per bucket, serialising the overlap)
        stage = torch.zeros(len(plist), dtype=plist[0].dtype)
        if flat.is_cuda:
            stage = stage.pin_memory()
        self.buckets.append(dict(flat=flat, nel=n, params=list(plist), pending=0, ready=False, handle=None, stage=stage))

    def zero_grad(self):
        """Zero the flat buffers and (re)attach the gradient views; call instead of optimizer.zero_grad()."""
        for b in self.buckets:
            b['flat'].zero_()
            off = 0
            for p in b['params']:
                p.grad = b['flat'][off:off + p.numel()].view_as(p)
                off += p.numel()
            b['pending'], b['ready'], b['handle'] = len(b['params']), False, None
        self._next = 0
        self._used = set()

    def _launch_ready(self):
        while self._next < len(self.buckets) and self.buckets[self._next]['ready']:
            b = self.buckets[self._next]
            # flags of the parameters that produced a gradient on this rank (world: > 0 after the averaged sum = some rank)
            # (the staging buffer is rewritten at most once per step and finish() ends every step with a host read, so the
            # previous step's copy has long completed)
            st = b['stage']
            for k, p in enumerate(b['params']):
                st[k] = float(self.world) if id(p) in self._used else 0.0
            b['flat'][b['nel']:].copy_(st, non_blocking=True)
            if self.world > 1:
                b['flat'].div_(self.world)
                b['handle'] = dist.all_reduce(b['flat'], async_op=True)
            self._next += 1

    def _on_grad(self, p):
        b = self.buckets[self._bucket_of[id(p)]]
        if p.grad.data_ptr() != b['flat'].data_ptr() + self._offset[id(p)][0] * p.element_size():   # autograd replaced the view: copy back
            self._reattach(b, p)
        self._used.add(id(p))
        b['pending'] -= 1
        if b['pending'] == 0:
            b['ready'] = True
            self._launch_ready()

    def _reattach(self, b, p):
        off = self._offset[id(p)][0]
        view = b['flat'][off:off + p.numel()].view_as(p)
        view.copy_(p.grad)
        p.grad = view

    def finish(self):
        """Issue the buckets backward did not complete (unused parameters), then wait for every collective."""
        for b in self.buckets:
            b['ready'] = True
        self._launch_ready()
        for b in self.buckets:
            if b['handle'] is not None:
                b['handle'].wait()
                b['handle'] = None
        # a parameter no rank produced a gradient for keeps grad = None, as after a plain backward (the optimizer then
        # skips it: no momentum / weight-decay step): the flags travelled with the buckets, one host read of all of them
        flags = torch.cat([b['flat'][b['nel']:].float() for b in self.buckets]).tolist()
        k = 0
        for b in self.buckets:
            for p in b['params']:
                if flags[k] == 0.0:
                    p.grad = None
                k += 1

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


class DistOptimizerHook(object):
    """zero_grad -> backward -> (all-reduce unless the model is DDP-wrapped) -> clip -> step.
    overlap=True: the all-reduce runs bucket by bucket during backward (OverlappedGradientReducer)."""

    def __init__(self, grad_clip=None, coalesce=True, bucket_size_mb=-1, ddp_wrapped=False, overlap=False, scaler=None):
        self.grad_clip = grad_clip
        self.coalesce = coalesce
        self.bucket_size_mb = bucket_size_mb
        self.ddp_wrapped = ddp_wrapped
        self.overlap = overlap
        self._reducer = None
        self.scaler = scaler          # torch.amp.GradScaler for fp16 autocast training (the reference's Fp16OptimizerHook role), or None

    def clip_grads(self, params):
        return clip_grad.clip_grad_norm_(filter(lambda p: p.requires_grad and p.grad is not None, params),
                                         **self.grad_clip)

    def after_train_iter(self, model, optimizer, loss):
        _, world = get_dist_info()
        if self.overlap and world > 1 and not self.ddp_wrapped:
            if self._reducer is None:
                self._reducer = OverlappedGradientReducer(model.parameters(),
                                                          self.bucket_size_mb if self.bucket_size_mb > 0 else 32)
            self._reducer.zero_grad()
            (self.scaler.scale(loss) if self.scaler is not None else loss).backward()
            self._reducer.finish()
            self._clip_and_step(model, optimizer)
            return
        optimizer.zero_grad()
        (self.scaler.scale(loss) if self.scaler is not None else loss).backward()
        if world > 1 and not self.ddp_wrapped:
            allreduce_grads(model.parameters(), self.coalesce, self.bucket_size_mb)
        self._clip_and_step(model, optimizer)

    def _clip_and_step(self, model, optimizer):
        if self.scaler is not None:
            self.scaler.unscale_(optimizer)               # clip on the true gradients; inf / nan steps are skipped by the scaler
        if self.grad_clip is not None:
            self.clip_grads(list(model.parameters()))
        if self.scaler is not None:
            self.scaler.step(optimizer)
            self.scaler.update()
        else:
            optimizer.step()


def parse_losses(losses):
    """Sum every `loss*` entry (lists are summed over levels) and all-reduce the logged scalars (train.py:35-56)."""
    log_vars = OrderedDict()
    for loss_name, loss_value in losses.items():
        if isinstance(loss_value, torch.Tensor):
            log_vars[loss_name] = loss_value.mean()
        elif isinstance(loss_value, (list, tuple)):
            log_vars[loss_name] = sum(_loss.mean() for _loss in loss_value)
        else:
            raise TypeError('{} is not a tensor or list of tensors'.format(loss_name))
    loss = sum(_value for _key, _value in log_vars.items() if 'loss' in _key)
    log_vars['loss'] = loss
    _, world = get_dist_info()
    for loss_name, loss_value in log_vars.items():
        v = loss_value.data.clone()
        if world > 1:
            dist.all_reduce(v.div_(world))
        log_vars[loss_name] = v.item()
    return loss, log_vars


def shard_indices(num_samples, rank, world_size, seed=0, epoch=0, samples_per_gpu=1):
    """The slice of a seeded permutation this rank processes; padded so that every rank gets the same count
    (a multiple of samples_per_gpu), as DistributedGroupSampler does."""
    g = torch.Generator()
    g.manual_seed(seed + epoch)
    perm = torch.randperm(num_samples, generator=g).tolist()
    per = -(-num_samples // (world_size * samples_per_gpu)) * samples_per_gpu
    total = per * world_size
    perm = (perm * (total // max(len(perm), 1) + 1))[:total]
    return perm[rank * per:(rank + 1) * per]


def train_step(model, optimizer, data, hook, autocast_dtype=None):
    """One iteration: forward_train -> parse_losses -> DistOptimizerHook (mmdet/apis/train.py:59-82).  autocast_dtype
    (torch.float16 / torch.bfloat16): the forward runs under torch.autocast -- library convolutions in that type, the
    hot-path operators keep fp32 arithmetic on fp32-cast inputs (custom_fwd), the loss is formed in fp32."""
    if autocast_dtype is not None:
        with torch.autocast(device_type='cuda', dtype=autocast_dtype):
            losses = model(**data)
        losses = {k: ([x.float() for x in v] if isinstance(v, (list, tuple)) else v.float()) for k, v in losses.items()}
    else:
        losses = model(**data)
    loss, log_vars = parse_losses(losses)
    hook.after_train_iter(model, optimizer, loss)
    return log_vars


def graph_backbone(model, sample_img):
    """Training with static patch shapes: capture the BACKBONE's forward and backward as two hipGraphs
    (`torch.cuda.make_graphed_callables`) so that an iteration replays them instead of launching the backbone's several hundred
    stock kernels from Python one by one -- the step is bound by the GPU, but the Python launch rate shows wherever the host has to
    wait for a count (the loss's positives) and then catch up.  The backbone is stock PyTorch-ROCm (ResNet: library convolutions,
    eval-mode BatchNorm, ReLU, max-pool), has no data-dependent control flow and no host-side caches; neck, head and loss stay eager
    (their weight packs are keyed on the parameters' version counters, and the loss has data-dependent shapes).  The parameters keep
    their identity: optimiser, gradient hooks and `state_dict` are unaffected.  Returns True when the backbone was captured.
    Measured (round 6, configs[2], 2 x 1024^2): 31.2 ms per step against 29.7 eager -- same losses and weights after three steps
    (tests/test_gpu_train_graph.py) but SLOWER: the replayed backward is one opaque node (no overlap with the head's backward on other
    streams, static input / output copies), and the 2.3 ms between wall and kernel-busy time of the eager step are 1 361 kernel
    boundaries at ~1.7 us each, not Python.  Off by default (`bench.py --graph-backbone 1`)."""
    if not (torch.cuda.is_available() and sample_img.is_cuda):
        return False
    was_training = model.backbone.training
    model.backbone = torch.cuda.make_graphed_callables(model.backbone, (sample_img.detach().clone(),))
    model.backbone.train(was_training)
    return True

