#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash tools/profile_round.sh <tag>
# kernel trace of the bench + the two PMC passes of the GAE kernel; outputs under gpurun_out/, summarised afterwards with
# tools/summarize_profiles.py <tag> into profiles/.
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline > $R/gpurun_out/prof_$tag.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_f -- python3 $R/tools/gae_once.py > $R/gpurun_out/pmc_${tag}_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_w -- python3 $R/tools/gae_once.py > $R/gpurun_out/pmc_${tag}_w.log 2>&1
find $R/gpurun_out -name "*.db" -delete
du -sh $R/gpurun_out
