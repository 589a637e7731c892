#!/bin/bash
# A/B builds of the update kernels for tools/train_only.py (ICRL_LIB=...): recompiles ppo_train_pairs.hip / ppo_train_rows.hip / ppo_train_halves.hip / ppo_train_quarters*.hip with
# extra -D flags and links them with the shipped objects of the other files.
#   bash tools/build_variant.sh fine -DICRL_FINE_PROF          -> icrl_amd/lib/var/libicrl_fine.so
# The shipped library (icrl_amd/lib/libicrl_hip.so) is never touched.
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/icrl_amd/lib/var
cd $R/icrl_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
for f in ppo_train_pairs ppo_train_rows ppo_train_halves ppo_train_quarters ppo_train_quarters2; do
  extra=""; [ $f = ppo_train_pairs ] && [ -z "$SLP" ] && extra="-fno-slp-vectorize"; [ $f = ppo_train_halves ] && [ -z "$SLP" ] && extra="-fno-slp-vectorize"      # as in the Makefile (SLP=1: with the vectoriser)
  /opt/rocm/bin/hipcc $FLAGS $extra "$@" -c $f.hip -o ../lib/var/${f}_$name.o &
done
wait
objs=""
for f in cn_train errors fine gae generic ppo_train rollout; do objs="$objs ../lib/obj/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/var/libicrl_$name.so $objs ../lib/var/ppo_train_pairs_$name.o ../lib/var/ppo_train_rows_$name.o ../lib/var/ppo_train_halves_$name.o ../lib/var/ppo_train_quarters_$name.o ../lib/var/ppo_train_quarters2_$name.o
echo built icrl_amd/lib/var/libicrl_$name.so
