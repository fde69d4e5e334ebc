"""CPU: the N > 1 path (image-parallel sharding + gradient all-reduce + max-over-ranks timing) with the gloo backend,
world_size 2, 127.0.0.1 rendezvous."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from orientedreppoints_amd import dist_utils as D
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    D.init_dist(backend='gloo')
    assert D.get_dist_info() == (rank, world)
    torch.manual_seed(0)                                   # same init on every rank
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 2, 1))
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    idx = D.shard_indices(10, rank, world, seed=3)
    data_all = torch.arange(10 * 3 * 8 * 8, dtype=torch.float32).reshape(10, 3, 8, 8) / 1000.0
    x = data_all[idx]

    class Wrapped(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, img):
            y = self.m(img)
            return {'loss_cls': y.pow(2).mean(), 'loss_rbox_init': [y.abs().mean(), y.mean().abs()]}
    w = Wrapped(model)
    hook = D.DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2))
    log = D.train_step(w, opt, dict(img=x), hook)
    # after the all-reduce every rank holds identical parameters
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    # bench-style timing: max over ranks
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, idx, log['loss'], bool(torch.equal(gathered[0], gathered[1])), float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_and_sharding():
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, idx0, loss0, same0, tmax0), (r1, idx1, loss1, same1, tmax1) = out
    assert len(idx0) == len(idx1) == 5 and sorted(idx0 + idx1) == list(range(10))     # disjoint cover
    assert same0 and same1                                                              # grads were averaged
    assert abs(loss0 - loss1) < 1e-12                                                   # logged loss is the rank mean
    assert tmax0 == tmax1 == 2.0


def _overlap_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from orientedreppoints_amd import dist_utils as D
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    D.init_dist(backend='gloo')

    class Net(torch.nn.Module):                            # a head that only some ranks use: different autograd graphs
        def __init__(self):
            super().__init__()
            self.stem = torch.nn.Conv2d(3, 8, 3, padding=1)
            self.body = torch.nn.Sequential(torch.nn.Conv2d(8, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 3, padding=1))
            self.head_a = torch.nn.Conv2d(8, 2, 1)
            self.head_b = torch.nn.Conv2d(8, 2, 1)
            self.unused = torch.nn.Parameter(torch.ones(5))

        def forward(self, img, use_b):
            f = self.body(torch.relu(self.stem(img)))
            loss = self.head_a(f).pow(2).mean()
            if use_b:
                loss = loss + self.head_b(f).abs().mean()
            return {'loss_cls': loss}

    results = {}
    # 'plain' / 'overlap': the same graphs on both ranks (the reference-style path needs that); 'ragged': overlap only,
    # rank 1 skips head_b in two of the three iterations
    for mode in ('plain', 'overlap', 'ragged'):
        torch.manual_seed(0)
        net = Net()
        opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)
        hook = D.DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2), overlap=(mode != 'plain'), bucket_size_mb=0.001)
        g = torch.Generator().manual_seed(100 + rank)
        for it in range(3):
            x = torch.randn(2, 3, 8, 8, generator=g)
            use_b = (it == 1) if mode != 'ragged' else (rank == 0 or it == 1)
            D.train_step(net, opt, dict(img=x, use_b=use_b), hook)
        results[mode] = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        if mode == 'overlap':
            nb = len(hook._reducer.buckets)
    same = True
    for mode in ('overlap', 'ragged'):
        flat = results[mode]
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        same = same and bool(torch.equal(gathered[0], gathered[1]))
    q.put((rank, float((results['plain'] - results['overlap']).abs().max()), same, nb))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_bucketed_allreduce_matches_the_plain_path():
    """OverlappedGradientReducer (async bucketed all-reduce from grad hooks, issued in a fixed order) == the reference-style
    all-reduce after backward, over three SGD steps, with a parameter that never gets a gradient and a head that only one
    rank uses in some iterations (the ranks' autograd graphs differ; the collectives must still match up)."""
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, diff, same, nb in out:
        assert diff <= 1e-6 and same and nb >= 3


def test_shard_indices_padding():
    from orientedreppoints_amd.dist_utils import shard_indices
    parts = [shard_indices(7, r, 4, seed=1, samples_per_gpu=2) for r in range(4)]
    assert all(len(p) == 2 for p in parts)
    assert set(sum(parts, [])) <= set(range(7)) and len(set(sum(parts, []))) == 7
