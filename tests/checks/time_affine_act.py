"""Dev script (GPU): the backbone's eval-BatchNorm (+ residual) + ReLU pass (csrc/orp_norm.hip affine_act_kernel) on the R-50 plane
sizes of a 1024^2 image, HIP-event timed; run once per library build (ORP_HIP_LIB=...) to compare two builds and check the outputs'
checksum is unchanged.   python tests/checks/time_affine_act.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from orientedreppoints_amd import _lib

SHAPES = [(64, 512, 1), (64, 256, 3), (256, 256, 3), (128, 256, 1), (128, 128, 4), (512, 128, 4), (256, 128, 1), (256, 64, 6),
          (1024, 64, 6), (512, 64, 1), (512, 32, 3), (2048, 32, 3)]          # (channels, side, calls per image)


def main():
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    L = _lib.lib()
    total = {False: 0.0, True: 0.0}
    h = hashlib.sha256()
    for C, S, n in SHAPES:
        x0 = torch.randn(1, C, S, S, device=dev)
        r = torch.randn(1, C, S, S, device=dev)
        a = torch.rand(C, device=dev) + 0.5
        b = torch.randn(C, device=dev)
        for with_res in (False, True):
            x = x0.clone()
            def run():
                rc = L.orp_affine_act(_lib.ptr(x), _lib.ptr(r) if with_res else None, _lib.ptr(a), _lib.ptr(b), _lib.ptr(x), 1, C,
                                      S * S, 1, _lib.stream_of(x))
                assert rc == 0
            run()
            h.update(x.cpu().numpy().tobytes())
            for _ in range(5):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / 50
            gb = C * S * S * 4 * (3 if with_res else 2) / 1e9
            print(f"C={C:5d} {S:3d}x{S:<3d} res={int(with_res)}  {us:7.1f} us  {gb / (us * 1e-6) / 1e3:5.2f} TB/s")
            total[with_res] += us * n
    print("weighted (calls per image):  plain %.0f us   residual %.0f us" % (total[False], total[True]))
    print("library", L.orp_version().decode(), " output sha256", h.hexdigest()[:16])


if __name__ == "__main__":
    main()
