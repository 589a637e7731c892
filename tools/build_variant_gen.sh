#!/bin/bash
# A/B builds of csrc/generic.hip (ICRL_HIP_LIB=...): recompiles it with extra -D flags and links it with the shipped objects of the other files.
#   bash tools/build_variant_gen.sh nopt -DGENP_X_NOPT          -> icrl_amd/lib/var/libicrl_nopt.so
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/icrl_amd/lib/var
cd $R/icrl_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS "$@" -c generic.hip -o ../lib/var/generic_$name.o
objs=""
for f in cn_train errors fine gae ppo_train ppo_train_pairs ppo_train_rows ppo_train_halves rollout; do objs="$objs ../lib/obj/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/var/libicrl_$name.so $objs ../lib/var/generic_$name.o
echo built icrl_amd/lib/var/libicrl_$name.so
