"""CPU, 2 gloo ranks: the REAL detector (ResNet-50 + FPN + OrientedRepPointsHead, its own loss() with the APAA assessment)
takes one `dist_utils.train_step` with the overlapped bucketed gradient all-reduce -- rank 0's image has ground truths, rank
1's has none, so the two ranks' autograd graphs and gradient arrival orders differ -- and must end with the parameters a
single process gets from the rank-averaged gradients (the reference's DDP semantics: per-rank loss normalisation,
mmdet/apis/train.py:137-141; head :441,455).  The HIP operators are replaced by oracle-backed CPU stand-ins
(tests/cpu_standins.py): what is under test is the host logic around them."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZE = 128


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(rank):
    from orientedreppoints_amd import synthetic as S
    g = torch.Generator().manual_seed(50 + rank)
    img = torch.randn(1, 3, SIZE, SIZE, generator=g)
    meta = [dict(img_shape=(SIZE, SIZE, 3), pad_shape=(SIZE, SIZE, 3), scale_factor=1.0, flip=False)]
    if rank == 0:
        gts = torch.from_numpy((S.gen_polys(2, 3, wh=(24, 60))[:, :8] * (SIZE / 1024.0) + 20).astype(np.float32))
        labels = torch.tensor([3, 11])
    else:
        gts, labels = torch.zeros((0, 8)), torch.zeros((0,), dtype=torch.long)
    return dict(img=img, img_meta=meta, gt_bboxes=[gts], gt_labels=[labels])


def _build():
    from orientedreppoints_amd.dota_configs import r50_model, test_cfg, train_cfg
    from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector
    torch.manual_seed(0)
    model = build_detector(ConfigDict(r50_model), train_cfg=ConfigDict(train_cfg), test_cfg=ConfigDict(test_cfg)).train()
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.01, momentum=0.9, weight_decay=1e-4)
    return model, opt


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    torch.set_num_threads(2)
    import cpu_standins
    from orientedreppoints_amd import dist_utils as D
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    D.init_dist(backend='gloo')
    with cpu_standins.installed():
        model, opt = _build()
        hook = D.DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2), overlap=True, bucket_size_mb=4)
        log = D.train_step(model, opt, _data(rank), hook)
        nbuckets = len(hook._reducer.buckets)
        mine = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        same = bool(torch.equal(gathered[0], gathered[1]))
        diff = used = None
        if rank == 0:
            # single process: per-rank gradients, averaged, clipped, one SGD step from the same initial state
            grads = []
            for r in range(world):
                ref, _ = _build()
                losses = ref(**_data(r))
                # parse_losses' total without its logging collective (the other rank is not taking part here)
                loss = sum(v.mean() if isinstance(v, torch.Tensor) else sum(x.mean() for x in v) for v in losses.values())
                loss.backward()
                grads.append([p.grad for p in ref.parameters()])
            ref, ropt = _build()
            used = 0
            for p, g0, g1 in zip(ref.parameters(), *grads):
                if g0 is None and g1 is None:
                    continue
                used += 1
                z = torch.zeros_like(p)
                p.grad = ((g0 if g0 is not None else z) + (g1 if g1 is not None else z)) / world
            torch.nn.utils.clip_grad_norm_([p for p in ref.parameters() if p.grad is not None], max_norm=35, norm_type=2)
            ropt.step()
            want = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
            diff = float((mine - want).abs().max())
    q.put((rank, same, diff, used, nbuckets, float(log['loss'])))
    dist.barrier()
    dist.destroy_process_group()


def test_detector_train_step_two_ranks_overlapped_reducer():
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (r0, same0, diff0, used, nb, loss0), (r1, same1, _, _, _, loss1) = out
    assert same0 and same1, "the ranks ended the step with different parameters"
    assert nb >= 2 and used > 150                       # several buckets; the whole detector took part
    assert diff0 <= 2e-6, "2-rank overlapped step differs from the single-process step on the averaged gradients: %g" % diff0
    assert abs(loss0 - loss1) < 1e-9                     # the logged loss is the mean over ranks
