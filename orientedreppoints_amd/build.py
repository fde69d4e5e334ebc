"""Compile the HIP kernels + C ABI into orientedreppoints_amd/csrc/liborp_hip.so (gfx950 only, in-tree).

hipcc cross-compiles without a GPU.  The geometry kernels are built with -ffp-contract=off because the fp32
operation order of the rotated-IoU core is part of the parity contract (bit-exact NMS decisions); the DeformConv
contraction is built with contraction on (MFMA / FMA is what it is for).
"""
import concurrent.futures
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "liborp_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

# -packed-fp32-ops: NO packed fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) in any kernel of this library.
# Measured on MI355X (tests/checks/mfma_refill_victim.hip, a self-contained reproducer): while other waves keep the
# matrix pipe busy with dense bf16 / f16 MFMAs, a v_pk_mul_f32 returns a wrong low half in lanes 48..63 -- 5.6e6 wrong results in
# 5e10, none for the same arithmetic as two v_mul_f32.  Every wrong-result anomaly of rounds 4 / 5 (wrong rows of the tile-height-1
# DeformConv launch, rows of the fp16-pieces build, a rotated-NMS keep count off by one next to another stream's convolutions) was
# this instruction in compiler-generated code; the scalar forms cost at most one VALU instruction per pair.
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-Wno-unused-result", "-Wno-unused-value",
          "-Wno-bitwise-instead-of-logical"]   # predicates are combined with & / | on purpose (no short-circuit branches)
if os.environ.get("ORP_PACKED_FP32", "0") != "1":   # ORP_PACKED_FP32=1: dev aid (build_variants with the packed forms)
    COMMON = COMMON + NO_PACKED_FP32
# (source, extra flags)
SOURCES = [
    ("orp_nms.hip", ["-ffp-contract=off"]),
    ("orp_overlaps.hip", ["-ffp-contract=off"]),
    ("orp_minarearect.hip", ["-ffp-contract=off"]),
    ("orp_box_iou_rotated.hip", ["-ffp-contract=off"]),
    ("orp_convex.hip", ["-ffp-contract=off"]),
    ("orp_convex_giou.hip", ["-ffp-contract=off"]),
    ("orp_pointwise.hip", ["-ffp-contract=off"]),
    ("orp_assign.hip", ["-ffp-contract=off"]),
    ("orp_postproc.hip", ["-ffp-contract=off"]),
    ("orp_soft_rnms.hip", ["-ffp-contract=off"]),
    ("orp_eval.hip", ["-ffp-contract=off"]),
    ("orp_train.hip", ["-ffp-contract=off"]),
    ("orp_norm.hip", []),
    ("orp_conv_small.hip", []),
    ("orp_conv1x1.hip", []),
    ("orp_dcn.hip", []),
    ("orp_dcn_split.hip", ["-ffp-contract=off"]),    # the bilinear combine is the reference's unfused float expression
    ("orp_conv_split.hip", []),
    ("orp_conv_wgrad.hip", []),
    ("orp_dcn_half.hip", []),
    ("orp_dcn_bwd.hip", []),
    ("orp_dcn_bwd_mfma.hip", []),
    ("orp_prof.hip", []),
]
HEADERS = ["orp_geom.hpp", "orp_quadfast.hpp", "orp_tile.hpp", "orp_hull.hpp", "orp_libm.hpp", "orp_prof.hpp", "orp_launch.hpp", "orp_dcn_split.hpp", os.path.join("..", "..", "include", "orp_hip.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


VERSION_SOURCE = "orp_overlaps.hip"      # defines orp_version(): rebuilt whenever any kernel source changes


def build_id():
    """12 hex digits over every kernel source / header (what `orp_version` reports after "abi1"): profiles/*_pmc.json
    record it, bench.py refuses to quote counters collected on another build."""
    import hashlib
    h = hashlib.sha1()
    names = sorted([s for s, _ in SOURCES] + [x for x in HEADERS])
    for n in names:
        path = os.path.join(CSRC, n)
        if os.path.exists(path):
            h.update(n.encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:12]


def _compile(src, extra):
    obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    if src == VERSION_SOURCE:
        deps += [os.path.join(CSRC, s) for s, _ in SOURCES]
        extra = extra + ['-DORP_BUILD_ID="%s"' % build_id()]
    if _stale(obj, deps):
        cmd = [HIPCC] + COMMON + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, universal_newlines=True)
        # (the host half of the compilation does not know the device feature of NO_PACKED_FP32 and says so once per function)
        err = "\n".join(l for l in r.stderr.splitlines() if "'-packed-fp32-ops' is not a recognized feature" not in l)
        if err.strip():
            import sys
            sys.stderr.write(err + "\n")
        if r.returncode != 0:
            raise subprocess.CalledProcessError(r.returncode, cmd)
    return obj


def build_hip(force=False, verbose=False):
    if not os.path.exists(HIPCC):
        raise RuntimeError("hipcc not found at %s: cannot build liborp_hip.so" % HIPCC)
    present = [(s, f) for s, f in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if force:
        for s, _ in present:
            o = os.path.join(CSRC, os.path.splitext(s)[0] + ".o")
            if os.path.exists(o):
                os.remove(o)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(present))) as ex:
        objs = list(ex.map(lambda sf: _compile(*sf), present))
    if _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build_hip(verbose=True))
