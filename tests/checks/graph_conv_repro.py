"""Development aid (GPU box): orp_conv_split_multi captured in a hipGraph and replayed on new data against the eager call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from orientedreppoints_amd.mmdet_ops.fused_norm import conv_split_multi, to_channels_last_multi, group_norm_act_multi_cl, Amax

dev = torch.device("cuda:0")
torch.manual_seed(0)
ca = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).to(dev)
cb = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).to(dev)
gn = torch.nn.GroupNorm(32, 256).to(dev)
# dirty the allocator's memory first
junk = [torch.full((1 << 20,), float('nan'), device=dev) for _ in range(64)]
del junk


def run(static, sizes, nprod, mode):
    xs = [s for s in static]
    if mode == 'prepass':
        cl = to_channels_last_multi(xs)
        a, b = conv_split_multi(cl, ca, cl, cb, nprod=nprod)
    else:
        cl, bits = to_channels_last_multi(xs, amax_slots=[0] * len(xs))
        a, b = conv_split_multi(cl, ca, cl, cb, nprod=nprod, amax=Amax(bits, 0))
    both, bits2 = group_norm_act_multi_cl(a + b, gn, relu=True, amax_slots=[0] * len(a) + [1] * len(b))
    a2, b2 = conv_split_multi(both[:len(a)], ca, both[len(a):], cb, nprod=nprod, amax=Amax(bits2, 1) if mode != 'prepass' else None)
    return [t.contiguous() for t in a2 + b2]


with torch.no_grad():
    for sizes in ((32, 16, 8), (32, 16, 8, 4, 2), (4, 2), (2,)):
        for B in (1, 2):
            for nprod in (3, 6):
                for mode in ('prepass', 'handover'):
                    static = [torch.randn(B, 256, n, n, device=dev) for n in sizes]
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        for _ in range(2):
                            run(static, sizes, nprod, mode)
                    torch.cuda.current_stream().wait_stream(side)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        outs = run(static, sizes, nprod, mode)
                    worst = 0.0
                    for seed in (1, 2, 3):
                        for s_ in static:
                            s_.copy_(torch.randn(s_.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(seed)) * (seed ** 2))
                        g.replay()
                        torch.cuda.synchronize()
                        got = [o.clone() for o in outs]
                        want = run(static, sizes, nprod, mode)
                        for u, v in zip(got, want):
                            d = float((u - v).abs().max()) if torch.isfinite(u).all() else float('inf')
                            worst = max(worst, d / max(1e-6, float(v.abs().max())))
                    print("levels %-18s B=%d nprod=%d %-8s: max |replay - eager| / scale = %.2e" % (sizes, B, nprod, mode, worst))
                    del g
