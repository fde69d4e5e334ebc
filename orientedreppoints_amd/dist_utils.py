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


class DistOptimizerHook(object):
    """zero_grad -> backward -> (all-reduce unless the model is DDP-wrapped) -> clip -> step."""

    def __init__(self, grad_clip=None, coalesce=True, bucket_size_mb=-1, ddp_wrapped=False):
        self.grad_clip = grad_clip
        self.coalesce = coalesce
        self.bucket_size_mb = bucket_size_mb
        self.ddp_wrapped = ddp_wrapped

    def clip_grads(self, params):
        return clip_grad.clip_grad_norm_(filter(lambda p: p.requires_grad and p.grad is not None, params),
                                         **self.grad_clip)

    def after_train_iter(self, model, optimizer, loss):
        optimizer.zero_grad()
        loss.backward()
        _, world = get_dist_info()
        if world > 1 and not self.ddp_wrapped:
            allreduce_grads(model.parameters(), self.coalesce, self.bucket_size_mb)
        if self.grad_clip is not None:
            self.clip_grads(list(model.parameters()))
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


def train_step(model, optimizer, data, hook):
    """One iteration: forward_train -> parse_losses -> DistOptimizerHook (mmdet/apis/train.py:59-82)."""
    losses = model(**data)
    loss, log_vars = parse_losses(losses)
    hook.after_train_iter(model, optimizer, loss)
    return log_vars
