"""Development aid (GPU box): the split DeformConv forward (and the same kernel as a convolution) at the configs[1] shapes under the ORP_DCNS_DBG timing switches (MODES=3,6,... selects the arithmetic modes)
(variant libraries from tools/build_variant.py; each in a fresh process):  python tests/checks/split_decomp.py d1,d2,d4,d8"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, os, ctypes
sys.path.insert(0, "%s")
import torch
from orientedreppoints_amd import _lib
from orientedreppoints_amd.mmdet_ops import deform_conv_forward_multi, deform_conv_forward_pair
dev = torch.device("cuda:0")
def prof(slot):
    tot = ctypes.c_double(0); cnt = ctypes.c_int(0)
    _lib.lib().orp_profile_read(slot, ctypes.cast(ctypes.byref(tot), ctypes.c_void_p), ctypes.cast(ctypes.byref(cnt), ctypes.c_void_p), 1)
    return tot.value / max(cnt.value, 1) * 1e3
torch.manual_seed(0)
IMG = int(os.environ.get("IMG", "1024")); B = int(os.environ.get("B", "1"))
w = torch.randn(256, 256, 3, 3, device=dev) * 0.01
w2 = torch.randn(256, 256, 3, 3, device=dev) * 0.01
xs = [torch.randn(B, 256, IMG // st, IMG // st, device=dev).contiguous(memory_format=torch.channels_last) for st in (8, 16, 32, 64, 128)]
xs2 = [torch.randn_like(x) for x in xs]
offs = [torch.randn(B, 18, IMG // st, IMG // st, device=dev) * 2 for st in (8, 16, 32, 64, 128)]
npos = B * sum((IMG // st) ** 2 for st in (8, 16, 32, 64, 128))
_lib.lib().orp_profile_enable(1)
out = []
for mode in [int(m) for m in os.environ.get("MODES", "0,6,9").split(",")]:
    _lib.lib().orp_dcn_set_split_mode(mode)
    for _ in range(3): deform_conv_forward_multi(xs, offs, w, 1, 1, 1, relu=True)
    torch.cuda.synchronize(); prof(3)
    for _ in range(20): deform_conv_forward_multi(xs, offs, w, 1, 1, 1, relu=True)
    torch.cuda.synchronize(); us1 = prof(3)
    for _ in range(3): deform_conv_forward_pair(xs, xs2, offs, w, w2, 1, 1, 1, relu=True)
    torch.cuda.synchronize(); prof(3)
    for _ in range(20): deform_conv_forward_pair(xs, xs2, offs, w, w2, 1, 1, 1, relu=True)
    torch.cuda.synchronize(); us2 = prof(3)
    # the same kernel without offsets (the towers' layer k of both towers): orp_conv_split_multi, slot 12
    from orientedreppoints_amd.mmdet_ops.fused_norm import conv_split_weights
    us3 = float("nan")
    if mode:
        for _ in range(3): conv_split_weights(xs, w, xs2, w2, nprod=mode)
        torch.cuda.synchronize(); prof(12)
        for _ in range(20): conv_split_weights(xs, w, xs2, w2, nprod=mode)
        torch.cuda.synchronize(); us3 = prof(12)
    out.append("mode %%d: single %%.1f us, pair %%.1f us (%%.1f TF/s fp32-equivalent), convolution pair %%.1f us" %% (mode, us1, us2, 4.0 * npos * 256 * 2304 / us2 / 1e6, us3))
print("  " + " | ".join(out))
''' % ROOT
names = sys.argv[1].split(',') if len(sys.argv) > 1 and sys.argv[1] else []
cfgs = [{}] + [{'ORP_HIP_LIB': os.path.join(ROOT, 'build_variants', 'liborp_hip_%s.so' % v)} for v in names]
for env in cfgs:
    print(env or 'as built', flush=True)
    e = dict(os.environ); e.update(env)
    subprocess.run([sys.executable, '-c', code], env=e)
