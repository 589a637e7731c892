"""SQ counters of the generic-shape persistent kernels (tools/pmc_generic.sh) -> profiles/<tag>_generic_pmc.md.
rocprofv3 sums a counter over all waves of the dispatch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_* are in quad-cycles (x 4 = shader cycles)."""
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
QUAD = {"SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
        "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_BUSY_CU_CYCLES"}
KERNELS = {"gen_train_persistent_kernel": ("optimiser step", 512), "rollout_generic_kernel": ("rollout step", 256)}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_generic_{tag}_[abc]"))):
    hits = sorted(glob.glob(os.path.join(d, "cc.csv")), key=os.path.getmtime)
    if not hits:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    t = {}
    for r in csv.DictReader(open(hits[-1])):
        name = next((k for k in KERNELS if k in r["Kernel_Name"]), None)
        if name is None:
            continue
        per[(name, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        t[(name, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    for (name, disp), cs in per.items():
        for c, v in cs.items():
            agg[name][c].append(v)
        dur[name].append(t[(name, disp)])
lines = [f"# {tag}: SQ counters of the generic-shape persistent kernels (`tools/pmc_generic.sh`, three `--pmc` passes over `tools/generic_only.py`, ONLY=4,5:",
         "`-sl 64 -pl 128 128 -rvl 64 -cvl 64 64 64`, 64 envs, batch 64)", "",
         "Per step and WAVE-AVERAGED where a counter is a per-wave sum (divide by the waves listed); cycles = quad-cycles x 4.", ""]
for name, (unit, steps) in KERNELS.items():
    if name not in agg:
        continue
    c = {k: sum(v) / len(v) for k, v in agg[name].items()}
    us = sum(dur[name]) / len(dur[name])
    lines += [f"## `{name}`: {len(dur[name])} dispatches, {us / 1e3:.2f} ms each under the counters = {us / steps:.1f} us per {unit} ({steps} steps per launch)", "",
              "| counter | per launch | per step |", "|---|---|---|"]
    for k in sorted(c):
        v = c[k] * (4 if k in QUAD else 1)
        lines.append(f"| {k}{' (cycles)' if k in QUAD else ''} | {v:.4g} | {v / steps:.4g} |")
    lines.append("")
open(os.path.join(ROOT, "profiles", f"{tag}_generic_pmc.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
