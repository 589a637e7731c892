#!/bin/bash
# A/B builds of ONE source file of csrc/ (default generic.hip; FILE=rollout ...): recompiles it with extra -D flags and links it with the shipped
# objects of the other files (ICRL_HIP_LIB=... selects the result).
#   bash tools/build_variant_gen.sh nopt -DGENP_X_NOPT          -> icrl_amd/lib/var/libicrl_nopt.so
set -e
name=$1; shift
file=${FILE:-generic}
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/icrl_amd/lib/var
cd $R/icrl_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $file.hip -o ../lib/var/${file}_$name.o
objs=""
for f in cn_train errors fine gae generic ppo_train ppo_train_pairs ppo_train_rows ppo_train_halves rollout; do [ $f = $file ] || objs="$objs ../lib/obj/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/var/libicrl_$name.so $objs ../lib/var/${file}_$name.o
echo built icrl_amd/lib/var/libicrl_$name.so
