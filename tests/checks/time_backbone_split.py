"""Dev script (GPU): would the R-50 backbone's convolutions run faster on this library's split matrix-pipe kernel (channels-last,
orp_conv_split_multi) than on the stock MIOpen / rocBLAS path they use now (NCHW F.conv2d)?  Every distinct (Cin, Cout, k, stride,
input side) of ResNet-50 at a 1024^2 image, HIP-event timed, weighted by calls per image.
   python tests/checks/time_backbone_split.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from orientedreppoints_amd.mmdet_ops.fused_norm import conv_split_weights

# (cin, cout, k, stride, input side, calls per image)      pytorch-style bottlenecks: the stride sits in the 3x3
SHAPES = [
    (64, 64, 1, 1, 256, 1), (64, 64, 3, 1, 256, 3), (64, 256, 1, 1, 256, 4), (256, 64, 1, 1, 256, 2),
    (256, 128, 1, 1, 256, 1), (128, 128, 3, 2, 256, 1), (128, 512, 1, 1, 128, 4), (256, 512, 1, 2, 256, 1),
    (512, 128, 1, 1, 128, 3), (128, 128, 3, 1, 128, 3),
    (512, 256, 1, 1, 128, 1), (256, 256, 3, 2, 128, 1), (256, 1024, 1, 1, 64, 6), (512, 1024, 1, 2, 128, 1),
    (1024, 256, 1, 1, 64, 5), (256, 256, 3, 1, 64, 5),
    (1024, 512, 1, 1, 64, 1), (512, 512, 3, 2, 64, 1), (512, 2048, 1, 1, 32, 3), (1024, 2048, 1, 2, 64, 1),
    (2048, 512, 1, 1, 32, 2), (512, 512, 3, 1, 32, 2),
]


def timed(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


def main():
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    tot = {"stock": 0.0, "split3": 0.0, "split6": 0.0}
    with torch.no_grad():
        for cin, cout, k, s, side, calls in SHAPES:
            x = torch.randn(1, cin, side, side, device=dev)
            w = torch.randn(cout, cin, k, k, device=dev) * (2.0 / (cin * k * k)) ** 0.5
            xcl = x.contiguous(memory_format=torch.channels_last)
            p = k // 2
            ref = F.conv2d(x, w, None, s, p)
            t_stock = timed(lambda: F.conv2d(x, w, None, s, p))
            row = [t_stock]
            for nprod in (3, 6):
                out = conv_split_weights([xcl], w, stride=(s, s), padding=(p, p), nprod=nprod)[0]
                err = float((out - ref).abs().max() / ref.abs().max())
                assert err < 1e-5, err
                row.append(timed(lambda: conv_split_weights([xcl], w, stride=(s, s), padding=(p, p), nprod=nprod)))
            gf = 2.0 * cin * cout * k * k * (side // s) ** 2 / 1e9
            mb = 4.0 * (cin * side * side / (s * s if k == 1 else 1) + cout * (side // s) ** 2) / 1e6
            print(f"{cin:5d}->{cout:<5d} {k}x{k} s{s} @{side:3d}  x{calls}  {gf:6.2f} GF {mb:6.1f} MB   stock {row[0]:7.1f} us   "
                  f"split3 {row[1]:7.1f}   split6 {row[2]:7.1f}")
            tot["stock"] += calls * row[0]; tot["split3"] += calls * row[1]; tot["split6"] += calls * row[2]
    print("per image: stock %.0f us   split (two fp16 pieces, with its range pre-pass) %.0f us   split (three bf16 pieces) %.0f us"
          % (tot["stock"], tot["split3"], tot["split6"]))


if __name__ == "__main__":
    main()
