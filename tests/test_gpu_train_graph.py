"""GPU (MI355X): training with the backbone's forward + backward replayed as two hipGraphs (dist_utils.graph_backbone; the
round-5 verdict's item on the training step's host gap) gives the eager step's losses and gradients, iteration after iteration,
with the optimiser's updates reaching the captured graphs (the parameters keep their identity)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _run(graph, steps=3):
    from orientedreppoints_amd import dist_utils as D, synthetic as S
    from orientedreppoints_amd.dota_configs import r50_model, test_cfg, train_cfg
    from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = build_detector(ConfigDict(r50_model), train_cfg=ConfigDict(train_cfg), test_cfg=ConfigDict(test_cfg)).to(dev).train()
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-3, momentum=0.9)
    hook = D.DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2), overlap=True, scaler=None)
    sz, B = 256, 2
    g = torch.Generator(device='cpu').manual_seed(7)
    data = dict(img=torch.randn(B, 3, sz, sz, generator=g).to(dev),
                img_meta=[dict(img_shape=(sz, sz, 3), pad_shape=(sz, sz, 3), scale_factor=1.0, flip=False)] * B,
                gt_bboxes=[torch.from_numpy(S.gen_polys(8, 40 + i, wh=(16, 120))[:, :8].astype(np.float32) * (sz / 1024.0)).to(dev) for i in range(B)],
                gt_labels=[torch.randint(1, 16, (8,), generator=g).to(dev) for _ in range(B)])
    if graph:
        assert D.graph_backbone(model, data['img'])
    losses = []
    for _ in range(steps):
        losses.append(float(D.train_step(model, opt, data, hook)['loss']))
    w = model.backbone.layer4[0].conv1.weight.detach().clone()
    w2 = model.bbox_head.reppoints_cls_out.weight.detach().clone()
    return losses, w, w2


def test_graphed_backbone_training_steps_match_eager():
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    le, we, he = _run(False)
    lg, wg, hg = _run(True)
    assert all(np.isfinite(le)) and le[0] != le[-1], "the optimiser must move the loss"
    for a, b in zip(le, lg):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(a)), (le, lg)
    # three SGD steps later the backbone and head weights agree: the replayed backward fed the optimiser the eager gradients
    assert float((we - wg).abs().max()) <= 1e-5 * float(we.abs().max())
    assert float((he - hg).abs().max()) <= 1e-5 * max(float(he.abs().max()), 1e-3)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_half_precision_detector_runs_end_to_end(dtype):
    """`model.half()` / `.bfloat16()` at inference: the head's DeformConvs take the half kernel (the reference's
    AT_DISPATCH_FLOATING_TYPES_AND_HALF branch), the post-processing widens the head's outputs to fp32 once.  Round 6: until then the
    pair launch narrowed nothing but returned fp32 into half 1x1 convolutions, and the fused post-processing handed half storage to
    kernels that take float pointers (a memory fault).  Checked: it runs, and finds the fp32 model's detection count to within 5 % (fp16) / 25 % (bf16)."""
    import copy
    from orientedreppoints_amd.dota_configs import r50_model, test_cfg
    from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = build_detector(ConfigDict(r50_model), train_cfg=None, test_cfg=ConfigDict(test_cfg)).to(dev).eval()
    head = model.bbox_head
    with torch.no_grad():
        head.reppoints_cls_out.weight.normal_(0, 0.05); head.reppoints_cls_out.bias.fill_(-3.3)
        head.reppoints_pts_init_out.bias.copy_(torch.tensor([[-1, -1], [-1, 0], [-1, 1], [0, -1], [0, 0], [0, 1], [1, -1], [1, 0], [1, 1]],
                                                            dtype=torch.float32, device=dev).reshape(-1) * 2.0)
    sz = 512
    metas = [dict(img_shape=(sz, sz, 3), pad_shape=(sz, sz, 3), scale_factor=1.0, flip=False)]
    img = torch.randn(1, 3, sz, sz, device=dev)
    with torch.no_grad():
        n32 = sum(len(c) for c in model.simple_test_batch(img, metas)[0])
        m = copy.deepcopy(model).to(dtype)
        res = m.simple_test_batch(img.to(dtype), metas)[0]
    n = sum(len(c) for c in res)
    tol = 0.05 if dtype == torch.float16 else 0.25          # (bf16: 8 significant bits next to a score threshold on a random-init model)
    assert n32 > 100 and abs(n - n32) <= tol * n32 + 5, (n, n32)
    assert all(np.isfinite(c).all() for c in res)
