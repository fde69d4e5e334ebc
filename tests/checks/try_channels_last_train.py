"""Development aid (GPU box): the training step with the BACKBONE in channels_last memory format (the library's NHWC implicit
GEMMs then skip their own layout conversions) next to the default NCHW, fp32 and fp16 autocast: time and loss."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from orientedreppoints_amd import synthetic as S, dist_utils as D
from orientedreppoints_amd.dota_configs import r50_model, train_cfg, test_cfg
from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector

dev = torch.device("cuda:0")
B, K = 2, 64


def run(channels_last, amp):
    torch.manual_seed(0)
    model = build_detector(ConfigDict(r50_model), train_cfg=ConfigDict(train_cfg), test_cfg=ConfigDict(test_cfg)).to(dev).train()
    if channels_last:
        model.backbone = model.backbone.to(memory_format=torch.channels_last)
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-4, momentum=0.9, weight_decay=1e-4)
    hook = D.DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2), overlap=True,
                               scaler=torch.amp.GradScaler('cuda') if amp == torch.float16 else None)
    g = torch.Generator(device='cpu').manual_seed(1234)
    img = torch.randn(B, 3, 1024, 1024, generator=g).to(dev)
    if channels_last:
        img = img.contiguous(memory_format=torch.channels_last)
    data = dict(img=img, img_meta=[dict(img_shape=(1024, 1024, 3), pad_shape=(1024, 1024, 3), scale_factor=1.0, flip=False)] * B,
                gt_bboxes=[torch.from_numpy(S.gen_polys(K, 40 + i, wh=(16, 120))[:, :8].astype(np.float32)).to(dev) for i in range(B)],
                gt_labels=[torch.randint(1, 16, (K,), generator=g).to(dev) for _ in range(B)])
    for _ in range(6):
        lv = D.train_step(model, opt, data, hook, autocast_dtype=amp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(15):
        lv = D.train_step(model, opt, data, hook, autocast_dtype=amp)
    torch.cuda.synchronize()
    print("backbone channels_last=%s autocast=%s: %.2f ms per step, loss %.4f" % (channels_last, amp, (time.perf_counter() - t0) / 15 * 1e3, float(lv['loss'])))


for amp in (None, torch.float16):
    for cl in (False, True):
        run(cl, amp)
