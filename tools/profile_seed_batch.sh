#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel trace of a lock-step batch of 32 runs; summarised by tools/summarize_profiles.py <tag>.
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export SEEDS=32
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_seeds -- python3 $R/tools/seed_batch_bench.py > $R/gpurun_out/prof_${tag}_seeds.log 2>&1
find $R/gpurun_out -name "*.db" -delete
grep "S=" $R/gpurun_out/prof_${tag}_seeds.log
