cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in base rep; do
  if [ $v = base ]; then L=""; else L=icrl_amd/lib/var/libicrl_$v.so; fi
  echo -n "$v: "; KIND=ant ICRL_LIB=$L VARIANTS=auto,auto,auto python tools/train_only.py 2>&1 | grep "us/step" | awk '{printf "%s ", $4}'; echo
done; done
timeout 600 python -m pytest tests/test_ppo_train_gpu.py -x -q -m gpu 2>&1 | tail -2
