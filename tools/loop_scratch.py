"""in-loop scratch traffic of an update kernel's ISA (tools/asm_dump.sh output):  python tools/loop_scratch.py kernel.s
prints, for the optimiser-step loop (the depth-1 loop that holds the MFMAs), the scratch loads / stores and v_accvgpr moves per barrier segment"""
import re, sys
L = open(sys.argv[1]).read().split("\n")
best = None
for h, l in enumerate(L):
    m = re.match(r"^\.(LBB\d+_\d+):\s*; =>This Loop Header: Depth=1", l)
    if not m:
        continue
    tag = "Header=" + m.group(1)[1:]
    members = [i for i, x in enumerate(L) if tag in x]
    if not members:
        continue
    lo, hi = min(h, min(members)), max(members)
    # the last member block runs until the next label
    while hi + 1 < len(L) and not re.match(r"^\.LBB\d+_\d+:", L[hi + 1]):
        hi += 1
    n = sum(1 for x in L[lo:hi] if "v_mfma" in x)
    if best is None or n > best[2]:
        best = (lo, hi, n, h)
lo, hi, n, h = best
print(f"loop lines {lo}..{hi} (header {h}): {hi - lo} lines, {n} MFMAs, scratch {sum('scratch_' in l for l in L[lo:hi])}")
bars = [h] + [i for i in range(h, hi) if "s_barrier" in L[i]] + [hi]
for a, b in zip(bars, bars[1:]):
    seg = L[a:b]
    ins = sum(1 for l in seg if l.startswith("\t") and not l.strip().startswith((";", ".")))
    print(f"  [{a}-{b}] instr {ins} scratch_load {sum('scratch_load' in l for l in seg)} scratch_store {sum('scratch_store' in l for l in seg)} "
          f"accvgpr {sum('v_accvgpr' in l for l in seg)} ds {sum(chr(9) + 'ds_' in l for l in seg)} mfma {sum('v_mfma' in l for l in seg)} branches {sum('s_cbranch' in l for l in seg)}")
