#!/bin/bash
# Marginal cost of each component of the update step on the critical path: builds ppo_train_pairs.hip with one component removed
# (-DICRL_DIAG=<bit>, wrong results) and times it.   bash tools/diag_train.sh build   (here)  |  bash tools/diag_train.sh run  (GPU box)
R=$(cd "$(dirname "$0")/.." && pwd)
bits="0 1 2 4 8 16 32 64 128 256 512 1024"
if [ "$1" = build ]; then
  for b in $bits; do bash $R/tools/build_variant.sh diag$b -DICRL_DIAG=$b > /dev/null 2>&1 & [ $((b % 3)) = 0 ] && wait; done; wait
  ls $R/icrl_amd/lib/var/ | grep -c diag
else
  cd $R
  names=(full -S5 -S6 -S7 -pair_handoffs -tanh -adam -loss -prefetch_staging -norm -adv_stats -book)
  i=0
  for b in $bits; do
    echo -n "${names[$i]}: "; ICRL_LIB=icrl_amd/lib/var/libicrl_diag$b.so VARIANTS=auto,auto python tools/train_only.py 2>&1 | grep "us/step" | awk '{printf "%s ", $4}'; echo; i=$((i+1))
  done
fi
