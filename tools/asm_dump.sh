#!/bin/bash
# ISA of one update-kernel instantiation:  bash tools/asm_dump.sh <file.hip> <mangled-name-substring> <out.s> [-D...]
f=$1; pat=$2; out=$3; shift 3
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -S --cuda-device-only "$@" $R/icrl_amd/csrc/$f -o $out.all 2>&1 | grep -v "warning\|^$" | head
awk -v pat="$pat" '$0 ~ "^_ZN4icrl.*"pat".*:" {f=1} f{print} /^\.Lfunc_end/{if(f) exit}' $out.all > $out
grep "NumVgprs\|ScratchSize\|TotalNumSgprs\|LDSByteSize\|Occupancy" $out | tr '\n' ' '; echo
echo "lines $(wc -l < $out) readlane $(grep -c v_readlane $out) writelane $(grep -c v_writelane $out) scratch $(grep -c scratch_ $out) mfma $(grep -c v_mfma $out)"
