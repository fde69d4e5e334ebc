"""Does MIOpen's fused conv + bias + ReLU (aten::miopen_convolution_relu / _add_relu) beat conv + a separate affine pass
for the backbone's layers?  (dev aid)"""
import torch
import torch.nn.functional as F


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device('cuda:0')
    cases = [  # cin, cout, k, stride, H
        (64, 64, 1, 1, 256), (64, 64, 3, 1, 256), (64, 256, 1, 1, 256), (256, 64, 1, 1, 256),
        (256, 128, 1, 1, 256), (128, 128, 3, 2, 256), (128, 512, 1, 1, 128), (512, 128, 1, 1, 128), (128, 128, 3, 1, 128),
        (512, 256, 1, 1, 128), (256, 256, 3, 2, 128), (256, 1024, 1, 1, 64), (1024, 256, 1, 1, 64), (256, 256, 3, 1, 64),
        (1024, 512, 1, 1, 64), (512, 512, 3, 2, 64), (512, 2048, 1, 1, 32), (2048, 512, 1, 1, 32), (512, 512, 3, 1, 32)]
    tot_a = tot_b = 0.0
    for cin, cout, k, s, H in cases:
        x = torch.randn(1, cin, H, H, device=dev)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        scale = torch.rand(cout, device=dev) + 0.5
        shift = torch.randn(cout, device=dev)
        wf = (w * scale.view(-1, 1, 1, 1)).contiguous()
        pad = k // 2

        def unfused():
            y = F.conv2d(x, w, None, s, pad)
            return torch.relu_(y.mul_(scale.view(1, -1, 1, 1)).add_(shift.view(1, -1, 1, 1)))

        def unfused1():                                   # one elementwise pass, like affine_act_kernel
            y = F.conv2d(x, w, None, s, pad)
            return torch.relu_(torch.addcmul(shift.view(1, -1, 1, 1), y, scale.view(1, -1, 1, 1), out=y))

        def conv_only():
            return F.conv2d(x, w, None, s, pad)

        def fused():
            return torch.ops.aten.miopen_convolution_relu(x, wf, shift, [s, s], [pad, pad], [1, 1], 1)
        try:
            a = fused()
            ok = float((a - unfused()).abs().max()) / max(1.0, float(a.abs().max()))
            tf = timed(fused)
        except Exception as e:  # noqa
            ok, tf = str(e)[:60], float('nan')
        tc, tu = timed(conv_only), timed(unfused1)
        tot_a += tu
        tot_b += tf
        print("cin %4d cout %4d k%d s%d H%3d: conv %6.1f  conv+affine %6.1f  fused %6.1f   relerr %s" % (cin, cout, k, s, H, tc, tu, tf, ok))
    print("sum conv+affine %.1f us   fused %.1f us" % (tot_a, tot_b))


if __name__ == '__main__':
    main()
