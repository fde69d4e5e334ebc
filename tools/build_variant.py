#!/usr/bin/env python3
"""Development aid: build liborp_hip.so variants with extra -D switches for A/B timing on the GPU box.
    python tools/build_variant.py <name> <source.hip>[,<source.hip>...] -DFOO=1 ...
compiles the named sources with the extra flags (every other object is taken from the regular in-tree build) and links
build_variants/liborp_hip_<name>.so; run a check script against it with ORP_HIP_LIB=build_variants/liborp_hip_<name>.so.
build_variants/ is git-ignored but travels to the GPU box."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orientedreppoints_amd import build as B  # noqa: E402


def main():
    name, srcs, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    if srcs == ["ALL"]:
        srcs = [s_ for s_, _ in B.SOURCES]
    B.build_hip()
    out_dir = os.path.join(ROOT, "build_variants")
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    for src, extra in B.SOURCES:
        base = os.path.splitext(src)[0]
        if src in srcs:
            obj = os.path.join(out_dir, "%s_%s.o" % (base, name))
            r = subprocess.run([B.HIPCC] + B.COMMON + extra + flags + ["-c", os.path.join(B.CSRC, src), "-o", obj], cwd=B.CSRC,
                               stderr=subprocess.PIPE, universal_newlines=True)
            sys.stderr.write("\n".join(l for l in r.stderr.splitlines() if "'-packed-fp32-ops' is not a recognized feature" not in l))
            if r.returncode != 0:
                raise SystemExit(r.returncode)
        else:
            obj = os.path.join(B.CSRC, base + ".o")
        objs.append(obj)
    lib = os.path.join(out_dir, "liborp_hip_%s.so" % name)
    subprocess.check_call([B.HIPCC, "--offload-arch=" + B.ARCH, "-shared", "-fPIC"] + objs + ["-o", lib])
    print(lib)


if __name__ == "__main__":
    main()
