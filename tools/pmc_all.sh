#!/bin/bash
# All rocprofv3 --pmc passes of the hot kernels (separate passes per counter set: FETCH_SIZE and WRITE_SIZE do not fit one
# pass; --kernel-trace only, no other trace domain) on tools/run_hot_kernels.py.  Run on the GPU box through gpurun:
#   gpurun -- 'bash tools/pmc_all.sh'      then locally:  python tools/make_pmc_json.py profiles/r0N gpurun_out/pmc_*
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -c "import orientedreppoints_amd._lib as L; print(L.lib().orp_version().decode())" > gpurun_out/pmc_version.txt
bash tools/pmc_pass.sh pmc_fetch all "FETCH_SIZE GRBM_GUI_ACTIVE" | tail -3
bash tools/pmc_pass.sh pmc_write all "WRITE_SIZE" | tail -3
bash tools/pmc_pass.sh pmc_sqa all "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" | tail -3
bash tools/pmc_pass.sh pmc_sqb all "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" | tail -3
cat gpurun_out/pmc_version.txt
