"""Turn the rocprofv3 outputs of tools/profile_round.sh (under gpurun_out/) into the tracked summaries under profiles/.

    python tools/summarize_profiles.py <tag>          e.g. r01_v2
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(pattern):
    # gpurun merges every call's files into gpurun_out/ (file names carry the profiled process id): take the newest
    hits = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", pattern), recursive=True), key=os.path.getmtime)
    return hits[-1] if hits else None


def kernel_stats(tag, which="bench"):
    sfx = "" if which == "bench" else "_antwall"
    src = find(f"prof_{tag}{sfx}/**/*kernel_stats.csv")
    if src is None:
        print("no kernel_stats.csv"); return
    rows = list(csv.DictReader(open(src)))
    dst = os.path.join(ROOT, "profiles", f"{tag}_{which}_kernel_stats")
    with open(dst + ".csv", "w") as f:
        f.write(open(src).read())
    bench, agree = "", ""
    log = os.path.join(ROOT, "gpurun_out", f"prof_{tag}{sfx}.log")
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("{\"metric\"") or line.startswith("{\"workload\""):
                j = json.loads(line)
                bench = (f"line of that (profiled) run: {j['value']:.0f} env-steps/s, {j['ms_per_step']:.0f} ms per outer iteration" +
                         (f", {j['us_per_optimizer_step']} us per optimiser step, {j['us_per_rollout_step']} us per rollout step" if which != "bench" else ""))
                if which == "bench" and "roofline" in j:
                    # the SAME process printed the line and was traced: its live event timing and the trace's average must agree
                    gae = [r for r in rows if "gae_dual_x4" in r["Name"]]
                    if gae:
                        avg_us = float(gae[0]["AverageNs"]) / 1e3
                        prof_frac = 2048 * 131072 * 36 / (avg_us * 1e-6) / 8e12
                        agree = (f"GAE roofline of the SAME run: the line says `roofline.achieved` {j['roofline']['achieved']} GB/s = {j['roofline']['frac']} of 8 TB/s "
                                 f"(HIP events inside bench.py); this trace's `gae_dual_x4_kernel` average is {avg_us:.1f} us = {2048 * 131072 * 36 / (avg_us * 1e-6) / 1e9:.0f} GB/s = "
                                 f"{prof_frac:.3f} ({100 * (prof_frac / j['roofline']['frac'] - 1):+.1f} %; the trace also holds the 3 warm-up launches of the sweep).  "
                                 f"Update kernel: the line says {j['roofline_ppo']['us_per_optimizer_step']} us per optimiser step.")
    # every launch of the update kernel, from the trace's timestamps, with the optimiser steps it ran at the line's time per step (the
    # target-KL early stop makes launches of one configuration differ in length: a launch is NOT classified by its duration.  The
    # placement calibration of rounds 3 / 4 — `PPOLagrangian.tune_sync_placement`, off by default since 0ecb14a — would show as 8
    # launches of <= 256 steps in front of the first update; it is only mentioned when the run switched it on: ICRL_TUNE_SYNC=1 in the log)
    upd = ""
    tr = find(f"prof_{tag}{sfx}/**/*kernel_trace.csv")
    if tr is not None:
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in csv.DictReader(open(tr)) if "ppo_train_" in r["Kernel_Name"] and "perm" not in r["Kernel_Name"] and "plan" not in r["Kernel_Name"]]
        if d:
            us = None
            if os.path.exists(log):
                for line in open(log):
                    if line.startswith("{\"metric\"") or line.startswith("{\"workload\""):
                        j = json.loads(line)
                        us = j.get("us_per_optimizer_step") or j.get("roofline_ppo", {}).get("us_per_optimizer_step")
            tuned = os.path.exists(log) and any("ICRL_TUNE_SYNC=1" in l for l in open(log))
            per = ", ".join(f"{x:.2f} ms" + (f" (~{round(x * 1e3 / us)} steps)" if us else "") for x in d)
            upd = (f"The {len(d)} launches of the update kernel in trace order (warm-up first): {per}" +
                   (f" — step counts at the line's {us} us per optimiser step; launches of one configuration differ in length where the target-KL test "
                    "(ppo_lag.py:293-297) ended their epoch loops early." if us else ".") +
                   (" The short launches in front of the first update are the placement calibration (`tune_sync_placement` was on in this run)." if tuned else ""))
    head = (f"# rocprofv3 --kernel-trace --stats — bench.py --steps 2 --warmup 1 ({tag})\n\n"
            "command: `rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_<tag> -- python3 bench.py --steps 2 "
            "--warmup 1 --no_cpu_baseline --no_seed_batch --no_configs2 --no_configs3 --no_configs4 --no_generic` (3 outer iterations traced incl. warm-up, plus the GAE sweep launches at N = 131 072)\n\n"
            if which == "bench" else
            f"# rocprofv3 --kernel-trace --stats — BASELINE configs[2] (AntWall-v0, 256 envs, batch 128, [40, 40]) ({tag})\n\n"
            "command: `rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_<tag>_antwall -- python3 tools/antwall_iter.py` "
            "(= bench.py's configs2 leg: 1 warm-up + 2 timed outer iterations)\n\n")
    with open(dst + ".md", "w") as f:
        f.write(head + f"{bench}\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows[:24]:
            name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-70:]
            f.write(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |\n")
        f.write("\n" + upd + "\n" + ("\n" + agree + "\n" if agree else ""))
    print("wrote", dst + ".md")


def seed_batch_stats(tag):
    """kernel table of a lock-step batch of 32 runs (tools/profile_seed_batch.sh)."""
    src = find(f"prof_{tag}_seeds/**/*kernel_stats.csv")
    if src is None:
        return
    rows = list(csv.DictReader(open(src)))
    dst = os.path.join(ROOT, "profiles", f"{tag}_seed_batch_kernel_stats")
    with open(dst + ".csv", "w") as f:
        f.write(open(src).read())
    line = ""
    log = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_seeds.log")
    if os.path.exists(log):
        line = "".join(l for l in open(log) if l.startswith("S="))
    with open(dst + ".md", "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats — 32 independent ICRL runs (BASELINE configs[1] each) in lock-step on one MI355X ({tag})\n\n"
                "command: `SEEDS=32 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_<tag>_seeds -- python3 tools/seed_batch_bench.py` "
                "(set-up + 1 warm-up + 2 timed outer iterations of every run; every `*_batch_kernel` launch carries all 32 runs, run = blockIdx.y)\n\n"
                f"the (profiled) run: {line}\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows[:20]:
            name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-70:]
            f.write(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |\n")
    print("wrote", dst + ".md")


def gae_pmc(tag, T=2048, N=131072, suffix=""):
    """suffix "": the streaming shape at 131 072 envs (what bench.py's roofline.traffic reads); "_mid": the register-resident split scan at 32 768."""
    import hashlib
    out = {}
    for name, key in (("FETCH_SIZE", "f"), ("WRITE_SIZE", "w")):
        src = find(f"pmc_{tag}{suffix}_{key}/**/*counter_collection.csv")
        if src is None:
            print("no counter csv for", name); return
        vals, keep = [], []
        for r in csv.DictReader(open(src)):
            if "gae_dual" in r["Kernel_Name"] and r["Counter_Name"] == name:
                vals.append(float(r["Counter_Value"])); keep.append(r)
        out[name] = vals
        with open(os.path.join(ROOT, "profiles", f"{tag}_gae{suffix}_pmc_{'fetch' if key == 'f' else 'write'}.csv"), "w") as f:
            w = csv.DictWriter(f, fieldnames=list(keep[0].keys())); w.writeheader(); w.writerows(keep)
        kname = keep[0]["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    fetch_kb, write_kb = max(out["FETCH_SIZE"]), max(out["WRITE_SIZE"])      # per dispatch (identical dispatches)
    alg = T * N * 36
    j = dict(kernel=kname, T=T, N=N, algorithmic_bytes=alg, FETCH_SIZE_kb=fetch_kb, WRITE_SIZE_kb=write_kb,
             fetch_bytes_raw=fetch_kb * 1024, fetch_bytes_corrected=fetch_kb * 1024 * 2, write_bytes=write_kb * 1024,
             traffic_bytes=fetch_kb * 1024 * 2 + write_kb * 1024,
             gae_hip_sha16=hashlib.sha256(open(os.path.join(ROOT, "icrl_amd", "csrc", "gae.hip"), "rb").read()).hexdigest()[:16],      # bench.py quotes the figure only for THIS source
             note="separate --pmc passes (FETCH_SIZE, WRITE_SIZE) as MI355X_MICROARCH.md §HBM prescribes; on gfx950 FETCH_SIZE counts "
                  "128-B requests at 64 B, so the read side is doubled; non-temporal loads / stores do not change the counts")
    with open(os.path.join(ROOT, "profiles", f"{tag}_gae{suffix}_pmc.json"), "w") as f:
        json.dump(j, f, indent=1)
    with open(os.path.join(ROOT, "profiles", f"{tag}_gae{suffix}_pmc.md"), "w") as f:
        f.write(f"# GAE kernel HBM traffic from PMC counters ({tag})\n\n"
                "    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_<tag>_f -- python3 tools/gae_once.py\n"
                "    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_<tag>_w -- python3 tools/gae_once.py\n\n"
                f"`{kname}`, T = {T}, N = {N}: algorithmic {alg} B (36 B x {T * N} transitions), per dispatch:\n\n"
                "| counter | raw KB | bytes | note |\n|---|---|---|---|\n"
                f"| FETCH_SIZE | {fetch_kb:.0f} | {fetch_kb * 1024:.0f} | x2 on gfx950 = {fetch_kb * 2048:.0f} B (algorithmic loads {T * N * 20} B) |\n"
                f"| WRITE_SIZE | {write_kb:.0f} | {write_kb * 1024:.0f} | algorithmic stores {T * N * 16} B |\n"
                f"| **traffic** | | **{j['traffic_bytes']:.0f}** | {100 * (j['traffic_bytes'] / alg - 1):+.2f} % vs algorithmic |\n")
    print("wrote", f"profiles/{tag}_gae{suffix}_pmc.json", j["traffic_bytes"] / alg)


if __name__ == "__main__":
    tag = sys.argv[1]
    kernel_stats(tag)
    kernel_stats(tag, "antwall")
    seed_batch_stats(tag)
    gae_pmc(tag)
    gae_pmc(tag, N=32768, suffix="_mid")
