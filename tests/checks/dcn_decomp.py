"""Development aid (GPU box): DeformConv forward timing at the BASELINE configs[1] shapes under the ORP_DCN_DBG /
ORP_DCN_MT switches (each configuration in a fresh process; the switches are read once)."""
import os, subprocess, sys
code = r'''
import sys, os
sys.path.insert(0, "%s")
import numpy as np, torch, ctypes
from orientedreppoints_amd import _lib
from orientedreppoints_amd.mmdet_ops import deform_conv_forward_multi
dev = torch.device("cuda:0")
def prof(slot):
    tot = ctypes.c_double(0); cnt = ctypes.c_int(0)
    _lib.lib().orp_profile_read(slot, ctypes.cast(ctypes.byref(tot), ctypes.c_void_p), ctypes.cast(ctypes.byref(cnt), ctypes.c_void_p), 1)
    return tot.value / max(cnt.value, 1) * 1e3
torch.manual_seed(0)
IMG = int(os.environ.get("IMG", "1024"))
w = torch.randn(256, 256, 3, 3, device=dev) * 0.01
xs = [torch.randn(1, 256, IMG // st, IMG // st, device=dev).contiguous(memory_format=torch.channels_last) for st in (8, 16, 32, 64, 128)]
offs = [torch.randn(1, 18, IMG // st, IMG // st, device=dev) * 2 for st in (8, 16, 32, 64, 128)]
for _ in range(3): deform_conv_forward_multi(xs, offs, w, 1, 1, 1)
torch.cuda.synchronize()
_lib.lib().orp_profile_enable(1); prof(3)
for _ in range(20): deform_conv_forward_multi(xs, offs, w, 1, 1, 1, relu=True)
torch.cuda.synchronize()
us = prof(3)
npos = sum((IMG // st) ** 2 for st in (8, 16, 32, 64, 128))
print("  dcn fwd %%d positions: %%.1f us  %%.1f TF/s  frac %%.3f" %% (npos, us, 2.0 * npos * 256 * 2304 / us / 1e6, 2.0 * npos * 256 * 2304 / us / 1e6 / 157.3))
for dt in (torch.float16, torch.bfloat16):
    hx = [x.to(dt) for x in xs]; ho = [o.to(dt) for o in offs]; hw = w.to(dt)
    for _ in range(3): deform_conv_forward_multi(hx, ho, hw, 1, 1, 1, relu=True)
    torch.cuda.synchronize(); prof(3)
    for _ in range(20): deform_conv_forward_multi(hx, ho, hw, 1, 1, 1, relu=True)
    torch.cuda.synchronize()
    ush = prof(3)
    print("  dcn fwd %%s: %%.1f us  %%.1f TF/s" %% (str(dt), ush, 2.0 * npos * 256 * 2304 / ush / 1e6))
from orientedreppoints_amd.mmdet_ops import deform_conv_forward_pair
xs2 = [torch.randn_like(x) for x in xs]
w2 = torch.randn(256, 256, 3, 3, device=dev) * 0.01
for _ in range(3): deform_conv_forward_pair(xs, xs2, offs, w, w2, 1, 1, 1, relu=True)
torch.cuda.synchronize(); prof(3)
for _ in range(20): deform_conv_forward_pair(xs, xs2, offs, w, w2, 1, 1, 1, relu=True)
torch.cuda.synchronize()
us2 = prof(3)
print("  dcn pair (both layers, one launch): %%.1f us = %%.1f us per layer  frac %%.3f" %% (us2, us2 / 2, 2 * 2.0 * npos * 256 * 2304 / us2 / 1e6 / 157.3))
''' % os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HERE = os.path.dirname(os.path.abspath(__file__))
cfgs = [{}] + ([{'ORP_HIP_LIB': os.path.join(HERE, 'liborp_dbg%s.so' % v)} for v in sys.argv[1].split(',')] if len(sys.argv) > 1 and sys.argv[1] else [])
cfgs += [{'ORP_DCN_GEN': v} for v in (sys.argv[2].split(',') if len(sys.argv) > 2 and sys.argv[2] else [])]
for env in cfgs:
    print(env, flush=True)
    e = dict(os.environ); e.update(env)
    subprocess.run([sys.executable, '-c', code], env=e)
