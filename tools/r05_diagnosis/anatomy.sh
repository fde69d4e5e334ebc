#!/bin/bash
# (GPU box) the ORP_DCNS_DBG timing variants of the split kernel: anatomy of a phase in the fp16-pieces mode
python tests/checks/anatomy_split.py 2>&1 | grep -v amdgpu.ids
for v in "$@"; do ORP_HIP_LIB=build_variants/liborp_hip_$v.so python tests/checks/anatomy_split.py 2>&1 | grep -v amdgpu.ids; done
