#!/usr/bin/env python3
"""Build oracle/_ref/libref_orp.so FROM THE REFERENCE'S OWN SOURCES (test infrastructure only).

TEST INFRASTRUCTURE -- never imported by the product package (orientedreppoints_amd/).

What it does
------------
The reference's hot-path math lives in ``__device__`` functions of CUDA ``.cu`` files that cannot be
built here (no nvcc, THC headers gone).  The ``__device__`` bodies are plain C++, so this recipe

  1. slices the *device-function line ranges* out of the files where they lie under /root/reference
     into a scratch dir under /tmp (never into the repo, never committed),
  2. wraps each slice in its own namespace with ``#define __device__`` / ``__global__`` / ``__shared__``
     stubs (see ``oracle/ref_shim.cpp``), and
  3. compiles with ``g++ -O2 -ffp-contract=off`` (no FMA contraction: this pins the fp32 operation order)
     into ``oracle/_ref/libref_orp.so``  (git-ignored; it does travel to the GPU box).

No reference source text is copied into the repository: the only artefact is the shared object.
If /root/reference is absent (GPU box) this script is a no-op and the prebuilt .so is used if present.

Line ranges (reference file:lines -> namespace) are listed in SLICES below and mirror SURVEY.md section 8c.
"""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("ORP_REFERENCE_ROOT", "/root/reference")
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "libref_orp.so")

# (slice name, reference-relative path, first line, last line)   [1-based, inclusive]
SLICES = [
    ("rnms_kernel",      "mmdet/ops/nms/src/rnms_kernel.cu",                    12, 147),
    ("rnms_cpu",         "mmdet/ops/nms/src/rnms_cpu.cpp",                       5, 163),
    ("poly_nms",         "DOTA_devkit/poly_nms_gpu/poly_nms_kernel.cu",         33, 212),
    ("poly_overlaps",    "DOTA_devkit/poly_nms_gpu/poly_overlaps_kernel.cu",    33, 328),
    ("minarearect",      "mmdet/ops/minarearect/src/minarearect_kernel.cu",     14, 452),
    ("convex_iou",       "mmdet/ops/iou/src/convex_iou_kernel.cu",              14, 295),
    ("convex_giou",      "mmdet/ops/iou/src/convex_giou_kernel.cu",             14, 804),
    ("points_justify",   "mmdet/ops/point_justify/src/points_justify_kernel.cu", 19, 102),
    ("focal",            "mmdet/ops/sigmoid_focal_loss/src/sigmoid_focal_loss_cuda.cu", 23, 97),
    ("chamfer",          "mmdet/ops/chamfer_2d/src/chamfer_2d.cu",              12, 124),
    ("dcn_im2col",       "mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu",        84, 243),
    ("dcn_col2im",       "mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu",       279, 335),
    ("dcn_col2im_coord", "mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu",       373, 436),
    # DCNv2: dmcn_im2col_bilinear / dmcn_get_gradient_weight / dmcn_get_coordinate_weight (:467-568) and the three
    # modulated kernels im2col (:570-633), col2im (:635-693), col2im_coord (:695-767)
    ("dcn_modulated",    "mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu",       467, 767),
]


def have_reference():
    return os.path.isdir(REF) and os.path.isfile(os.path.join(REF, SLICES[0][1]))


def build(verbose=False):
    if not have_reference():
        if verbose:
            print("[oracle/_ref] %s not present: skipping (prebuilt .so %s)" %
                  (REF, "found" if os.path.exists(OUT) else "absent"))
        return os.path.exists(OUT)
    os.makedirs(OUT_DIR, exist_ok=True)
    shim = os.path.join(HERE, "ref_shim.cpp")
    deps = [shim, os.path.abspath(__file__)] + [os.path.join(REF, s[1]) for s in SLICES]
    if os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return True
    with tempfile.TemporaryDirectory(prefix="orp_ref_") as tmp:
        for name, rel, lo, hi in SLICES:
            with open(os.path.join(REF, rel), "r", errors="replace") as f:
                lines = f.readlines()
            with open(os.path.join(tmp, name + ".inc"), "w") as g:
                g.writelines(lines[lo - 1:hi])
        cmd = ["g++", "-O2", "-std=c++14", "-ffp-contract=off", "-fPIC", "-shared", "-w",
               "-I", tmp, "-I", REF,
               "-DORP_REF_ROOT=\"%s\"" % REF,
               shim, "-o", OUT, "-lm"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return True


if __name__ == "__main__":
    ok = build(verbose=True)
    sys.exit(0 if ok else 1)
