#!/bin/bash
# Run ON THE GPU BOX (through gpurun): instruction-cache counters of the rollout kernels (tools/rollout_only.py, MODES / CFGS from the env).
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_INST_ANY\|SQ_INSTS_[A-Z_]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/pmc_icache_${tag}_avail.txt
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_icache_${tag} -- python3 $R/tools/rollout_only.py > $R/gpurun_out/pmc_icache_${tag}.log 2>&1
find $R/gpurun_out -name "*.db" -delete
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$R/gpurun_out/pmc_icache_${tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    if "rollout" in k:
        print(k, {n: int(x) for n, x in v.items()})
PY
