"""SQ counters of the PPO-Lagrangian update kernels (tools/pmc_train.sh) -> profiles/<tag>_train_pmc.md.

    python tools/summarize_pmc_train.py r02
Per launch of tools/train_only.py: 2 epochs x 2048 minibatches = 4096 optimiser steps, 3 workgroups.  rocprofv3 sums a counter
over all waves of the dispatch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are in quad-cycles (x 4 = shader cycles)."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
kind = sys.argv[2] if len(sys.argv) > 2 else "hc"           # "ant": tools/train_only.py with KIND=ant (64 envs x 512 rows, batch 128)
STEPS = 4096 if kind == "hc" else 512
suffix = "" if kind == "hc" else "_antwall"
QUAD = {"SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
        "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_BUSY_CU_CYCLES"}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
waves, dur = {}, {}
kept, dropped = collections.Counter(), collections.Counter()
clocks, us_step = collections.defaultdict(list), collections.defaultdict(list)
# one file per pass directory (pmc_train_<tag>_a / _b): gpurun merges every call's files into gpurun_out/, take the newest of each
_passes = []
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_train_{tag}{suffix}_[abc]"))):
    if os.path.isdir(d):
        hits = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
        if hits:
            _passes.append(hits[-1])
for f in _passes:
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ppo_train_" not in k or "perm" in k or "plan" in k:
            continue
        # with its template arguments: the SPLIT argument tells the two launch shapes of the row-owning kernel apart, the last argument of the
        # wave-quad kernel its two (round 5) and four (round 6) workgroups per network
        name = k.split("(")[0].split("::")[-1].replace("void ", "")
        per[(name, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[(name, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        # the WORKING waves: the update launches are 1-D grids of 8 (M - 1) + 1 workgroups of which M = 3 or 6 (the run's, at b, b + 8,
        # ...: one XCD) do the work and the others leave at once (their cycles are noise in the sums)
        wgs = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
        wgs = (wgs - 1) // 8 + 1 if wgs > 12 else wgs
        waves[name] = int(r["Workgroup_Size"]) // 64 * wgs
    # Only the FULL launches (STEPS optimiser steps) count: every process also makes 8 short calibration launches of the same kernel
    # (PPOLagrangian._tune_sync_placement: 256 steps each at HC), and a median over all launches divided by STEPS would describe those
    # (round 3's tables did: 1 376 "cycles per wave-step" for a kernel that takes 21 334).  A full launch is one that lasts at least
    # half as long as the kernel's longest.
    longest = collections.defaultdict(float)
    for (name, d_), t in dur.items():
        longest[name] = max(longest[name], t)
    for (name, d_), c in per.items():
        if dur[(name, d_)] < 0.5 * longest[name]:
            dropped[name] += 1
            continue
        kept[name] += 1
        for cn, v in c.items():
            agg[name][cn].append(v)
        if "SQ_WAVE_CYCLES" in c:      # cross-check: wave cycles per wave must be the launch's duration at a plausible shader clock
            ghz = c["SQ_WAVE_CYCLES"] * 4 / waves[name] / dur[(name, d_)] * 1e-9
            assert 1.2 < ghz < 2.7, f"{name} dispatch {d_}: SQ_WAVE_CYCLES x 4 / waves / duration = {ghz:.2f} GHz — not a shader clock"
            clocks[name].append(ghz)
            us_step[name].append(dur[(name, d_)] / STEPS * 1e6)
shape = "HCWithPos shapes, batch 64" if kind == "hc" else "AntWall shapes (obs 113, act 8), batch 128 = two 64-row chunks"
lines = [f"# SQ counters of the PPO-Lagrangian update kernels ({tag}) — {shape}, per WAVE and optimiser step",
         "", "command: `bash tools/pmc_train.sh <tag>` on the GPU box = two `rocprofv3 --pmc <8 SQ counters> --kernel-trace` passes over "
         "`tools/train_only.py` (HC, round 6: VARIANTS=halves,auto = the wave-quad kernel with TWO workgroups per network, `<2, false, 18, 2>`: 48 waves on 6 CUs, "
         "and with FOUR, `<2, false, 18, 4>`, the default: 96 waves on 12 CUs, 4096 optimiser steps per launch; until round 5 the columns were the wave-pair and the two-workgroup kernel; KIND=ant, round 6: VARIANTS=rows,auto = the row-owning kernel with two workgroups per network, 24 waves on 6 CUs, and "
         "the default `ppo_train_quarters2_kernel`: four workgroups per network, 48 waves on 12 CUs, 512 steps per launch; until round 5 the columns were one and two workgroups per network); raw csv: gpurun_out/pmc_train_<tag>_{a,b} (scratch).  Values "
         f"below = median over the FULL launches ({STEPS} optimiser steps; the short sync-placement calibration launches every process makes are dropped by duration) "
         f"/ waves of the kernel / {STEPS}; cycle-type counters converted from quad-cycles to shader cycles.", ""]
names = sorted(agg)
if kind != "hc":      # the same kernel symbol runs with 3 and with 6 workgroups: keep them apart by wave count
    pass
cols = ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_VALU", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_SALU",
        "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_SMEM",
        "SQ_INSTS_VMEM_RD", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_FMA_F32",
        "SQ_INSTS_BRANCH", "SQ_ACTIVE_INST_MISC", "SQ_BUSY_CU_CYCLES"]
lines.append("| counter | " + " | ".join(f"`{n}` ({waves[n]} waves)" for n in names) + " |")
lines.append("|---|" + "---|" * len(names))
lines.append("| full launches used (calibration launches dropped) | " + " | ".join(f"{kept[n]} ({dropped[n]})" for n in names) + " |")
lines.append("| us per optimiser step (dispatch timestamps, profiled) | " + " | ".join(f"{sorted(us_step[n])[len(us_step[n]) // 2]:.2f}" if us_step[n] else "-" for n in names) + " |")
lines.append("| implied shader clock, GHz (WAVE_CYCLES x 4 / waves / duration) | " + " | ".join(f"{sorted(clocks[n])[len(clocks[n]) // 2]:.2f}" if clocks[n] else "-" for n in names) + " |")
for c in cols:
    row = []
    for n in names:
        v = agg[n].get(c)
        if not v:
            row.append("-"); continue
        med = sorted(v)[len(v) // 2] / waves[n] / STEPS * (4 if c in QUAD else 1)
        row.append(f"{med:,.0f}")
    lines.append(f"| {c}{' (cycles)' if c in QUAD else ''} | " + " | ".join(row) + " |")
if kind != "hc":
    lines += ["", "`ppo_train_rows_kernel<8, false, true>` = rounds 3-5's default at batch 128 (`train_kernel = \"rows\"`): TWO workgroups per network (24 waves on 6 CUs), "
              "each wave carries 16 rows of one 64-row chunk through all features; `ppo_train_quarters2_kernel<8, false, 113>` = round 6's default: FOUR workgroups per "
              "network (48 waves on 12 CUs), part p takes rows 16 p .. of both chunks as two row tiles per wave of a wave quad, K = 32 weight-gradient GEMMs, "
              "the four partial gradients summed as (own + partner) + (the other pair) by all four.  Per wave and step a quarter of the MFMAs of one "
              "workgroup per network; the loss tail and Adam are replicated (VALU per wave does not drop), the exchange pulls three peers' blocks."]
    open(os.path.join(ROOT, "profiles", f"{tag}_train_pmc{suffix}.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    raise SystemExit(0)
lines += ["", "Reading: SQ_WAVE_CYCLES = ACTIVE_INST_ANY (issuing) + WAIT_INST_ANY (issue stalled: here the SIMD's one fp32 lane array, which the fp32 MFMA "
          "occupies alone: SQ_VALU_MFMA_COEXEC_CYCLES = 0, MFMA_BUSY = 32 cycles x SQ_INSTS_MFMA) + WAIT_ANY (parked at s_waitcnt / s_barrier).  "
          "Four workgroups per network (`<.., 4>`, 16 rows = one row tile each): the per-wave figures average over waves 0..3, which run forward / loss / "
          "backward alone on their SIMDs, and waves 4..7, which only take part in the weight-gradient GEMMs, the exchange and Adam (and are parked meanwhile); "
          "the MFMAs per wave halve again, the VALU work of the loss tail and of Adam does not (all four parts run them on their replicas), and the exchange "
          "now pulls three partners' blocks through the compute unit's L2 port (SQ_INSTS_VMEM_RD)."]
open(os.path.join(ROOT, "profiles", f"{tag}_train_pmc{suffix}.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
