#!/bin/bash
# Run ON THE GPU BOX (through gpurun): SQ counters of the PPO-Lagrangian update kernels, three passes of <= 8 SQ counters each.
#   bash tools/pmc_train.sh <tag>            HC shapes: the wave-quad kernel with two workgroups per network (48 waves on 6 CUs) and with four (the default: 96 waves on 12 CUs)
#   bash tools/pmc_train.sh <tag> ant        AntWall shapes, batch 128: the row-owning kernel with two workgroups per network (rows: rounds 3-5's default) and four workgroups per network, both chunks in one pass (round 6's default)
# Outputs under gpurun_out/pmc_train_<tag>[_antwall]_{a,b}; summarised by tools/summarize_pmc_train.py <tag> [ant].
tag=$1
kind=${2:-hc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
sfx=""; variants="halves,auto"
if [ "$kind" = "ant" ]; then sfx="_antwall"; variants="rows,auto"; fi
cd /tmp && export TMPDIR=/tmp
export KIND=$kind VARIANTS=$variants EPOCHS=2
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_train_${tag}${sfx}_a -- python3 $R/tools/train_only.py > $R/gpurun_out/pmc_train_${tag}${sfx}_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/pmc_train_${tag}${sfx}_b -- python3 $R/tools/train_only.py > $R/gpurun_out/pmc_train_${tag}${sfx}_b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_FMA_F32 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_train_${tag}${sfx}_c -- python3 $R/tools/train_only.py > $R/gpurun_out/pmc_train_${tag}${sfx}_c.log 2>&1
find $R/gpurun_out -name "*.db" -delete
ls -R $R/gpurun_out/pmc_train_${tag}${sfx}_a | head
