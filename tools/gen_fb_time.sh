#!/bin/bash
# per-kernel time of the generic-shape path under rocprofv3 for one library build: tools/generic_only.py, kernel stats
#   bash tools/gen_fb_time.sh <path to libicrl_*.so> <tag>        (on the GPU box; output under gpurun_out/prof_gen_<tag>)
lib=$1; tag=$2
cd /tmp && export TMPDIR=/tmp
ICRL_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_gen_$tag -- python3 $GRAFT_REPO_ROOT/tools/generic_only.py > /dev/null 2>&1
f=$(ls $GRAFT_REPO_ROOT/gpurun_out/prof_gen_$tag/*/*kernel_stats.csv | head -1)
echo "== $tag"
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("gen_", "policy_generic", "act_step_generic", "cn_cost_rows")):
        print(f'{r["Name"].split("(")[0]:48s} calls {r["Calls"]:>6s}  avg {float(r["AverageNs"]) / 1e3:7.1f} us  min {float(r["MinNs"]) / 1e3:7.1f}  max {float(r["MaxNs"]) / 1e3:7.1f}')
PY
