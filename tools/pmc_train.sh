#!/bin/bash
# Run ON THE GPU BOX (through gpurun): SQ counters of the PPO-Lagrangian update kernels (rows = one wave per SIMD, pairs = two),
# two passes of <= 8 SQ counters each.  Outputs under gpurun_out/pmc_train_<tag>_{a,b}; summarised by tools/summarize_pmc_train.py.
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
VARIANTS=rows,auto EPOCHS=2 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_train_${tag}_a -- python3 $R/tools/train_only.py > $R/gpurun_out/pmc_train_${tag}_a.log 2>&1
VARIANTS=rows,auto EPOCHS=2 timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/pmc_train_${tag}_b -- python3 $R/tools/train_only.py > $R/gpurun_out/pmc_train_${tag}_b.log 2>&1
find $R/gpurun_out -name "*.db" -delete
ls -R $R/gpurun_out/pmc_train_${tag}_a | head
