#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash tools/profile_generic.sh <tag>
# kernel trace of the generic-shape leg (tools/generic_only.py: the bench's generic_shape workload and the 128-wide one) -> gpurun_out/prof_<tag>_generic;
# summarised by `python tools/summarize_generic.py <tag>` into profiles/<tag>_generic_kernel_stats.{csv,md}
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
ONLY=${ONLY:-4,5} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_generic -- python3 $R/tools/generic_only.py > $R/gpurun_out/prof_${tag}_generic.log 2>&1
find $R/gpurun_out -name "*.db" -delete
tail -3 $R/gpurun_out/prof_${tag}_generic.log
