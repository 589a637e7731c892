#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash tools/profile_round.sh <tag>
# kernel trace of the bench + the two PMC passes of the GAE kernel + SQ counters of the update kernels + rollout / seed-batch
# tables; outputs under gpurun_out/, summarised afterwards with tools/summarize_profiles.py <tag> and
# tools/summarize_pmc_train.py <tag> into profiles/.
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_seed_batch --no_configs2 --no_configs3 --no_configs4 --no_generic > $R/gpurun_out/prof_$tag.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_f -- python3 $R/tools/gae_once.py > $R/gpurun_out/pmc_${tag}_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_w -- python3 $R/tools/gae_once.py > $R/gpurun_out/pmc_${tag}_w.log 2>&1
export GAE_N=32768      # the register-resident split scan (mid range): does it read every byte once?
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_mid_f -- python3 $R/tools/gae_once.py > $R/gpurun_out/pmc_${tag}_mid_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_mid_w -- python3 $R/tools/gae_once.py > $R/gpurun_out/pmc_${tag}_mid_w.log 2>&1
unset GAE_N
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_antwall -- python3 $R/tools/antwall_iter.py > $R/gpurun_out/prof_${tag}_antwall.log 2>&1
cd $R
bash tools/pmc_train.sh $tag > /dev/null 2>&1
bash tools/pmc_train.sh $tag ant > /dev/null 2>&1
timeout 300 python3 tools/rollout_only.py > gpurun_out/rollout_$tag.log 2>&1
VARIANTS=pairs,halves,auto,pairs,halves,auto timeout 300 python3 tools/train_only.py > gpurun_out/train_$tag.log 2>&1
KIND=ant timeout 300 python3 tools/train_only.py >> gpurun_out/train_$tag.log 2>&1
timeout 600 python3 tools/seed_batch_bench.py > gpurun_out/seeds_$tag.log 2>&1
bash tools/profile_seed_batch.sh $tag > /dev/null 2>&1
find $R/gpurun_out -name "*.db" -delete
du -sh $R/gpurun_out
