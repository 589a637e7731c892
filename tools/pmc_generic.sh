#!/bin/bash
# Run ON THE GPU BOX (through gpurun): SQ counters of the generic-shape persistent kernels (gen_train_persistent_kernel, rollout_generic_kernel),
# three passes of <= 8 SQ counters each over tools/generic_only.py (ONLY=4,5: the bench's generic_shape workload).
#   bash tools/pmc_generic.sh <tag>   -> gpurun_out/pmc_generic_<tag>_{a,b,c}; summarised by tools/summarize_pmc_generic.py <tag>
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export ONLY=${ONLY:-4,5}
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_generic_${tag}_a -- python3 $R/tools/generic_only.py > $R/gpurun_out/pmc_generic_${tag}_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/pmc_generic_${tag}_b -- python3 $R/tools/generic_only.py > $R/gpurun_out/pmc_generic_${tag}_b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_FMA_F32 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_generic_${tag}_c -- python3 $R/tools/generic_only.py > $R/gpurun_out/pmc_generic_${tag}_c.log 2>&1
find $R/gpurun_out -name "*.db" -delete
for p in a b c; do      # keep the rows of the two kernels only (the full counter files of a torch process are ~35 MB each)
  d=$R/gpurun_out/pmc_generic_${tag}_$p
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && { head -1 $f > $d/cc.csv; grep -E "gen_train_persistent_kernel|rollout_generic_kernel" $f >> $d/cc.csv; }
  find $d -name "*.csv" ! -name cc.csv -delete
done
ls $R/gpurun_out | grep pmc_generic
