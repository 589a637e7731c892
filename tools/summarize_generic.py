"""profiles/<tag>_generic_kernel_stats.{csv,md} from the rocprofv3 output of tools/profile_generic.sh (gpurun_out/prof_<tag>_generic)."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
hits = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_generic", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(hits[-1])))
dst = os.path.join(ROOT, "profiles", f"{tag}_generic_kernel_stats")
open(dst + ".csv", "w").write(open(hits[-1]).read())
log = [l.strip() for l in open(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_generic.log")) if "us per optimiser step" in l]
total = sum(float(r["TotalDurationNs"]) for r in rows)
with open(dst + ".md", "w") as f:
    f.write(f"# {tag}: kernel trace of the generic-shape leg (`rocprofv3 --kernel-trace --stats -- python3 tools/generic_only.py`, ONLY=4,5: the bench's `generic_shape` workload)\n\n")
    for l in log:
        f.write(f"line of that (profiled) run: {l}\n\n")
    f.write("| kernel | calls | total ms | average us | % of traced time |\n|---|---|---|---|---|\n")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
        f.write(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {100 * float(r['TotalDurationNs']) / total:.1f} |\n")
    f.write("\n`gen_train_persistent_kernel`: one launch per `train()` (2 epochs x 256 minibatches of 64 = 512 optimiser steps); `rollout_generic_kernel`: one launch per "
            "`collect_rollouts()` (256 steps of 64 envs); the tool also runs the Python loop over the fine-grained entry points once (`policy_generic_kernel`, "
            "`act_step_generic_kernel`, the normaliser: 4 launches per step) as the bit-identity reference.\n")
print(open(dst + ".md").read())
