#!/bin/bash
# Run ON THE GPU BOX: us per optimiser step of several A/B builds of the update kernels (tools/build_variant.sh), interleaved
#   bash tools/ab_train.sh ep0 ep1 [...]          (KIND=ant for AntWall shapes; REPS=2 rounds)
cd ${GRAFT_REPO_ROOT:-.}
for rep in $(seq ${REPS:-2}); do
  for v in "$@"; do
    echo -n "$v: "; ICRL_LIB=icrl_amd/lib/var/libicrl_$v.so VARIANTS=${VARIANTS:-auto,auto,auto} python tools/train_only.py 2>&1 | grep "us/step" | awk '{printf "%s ", $4}'; echo
  done
done
