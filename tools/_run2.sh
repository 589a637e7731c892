cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ppo_train_gpu.py -x -q -m gpu -k "ant or layout or two_chunk" 2>&1 | tail -5 > gpurun_out/r05_t2.log
for rep in 1 2; do
for v in base raw0; do
  if [ $v = base ]; then L=""; else L=icrl_amd/lib/var/libicrl_$v.so; fi
  echo "== $v" >> gpurun_out/r05_train_ant1.log
  ICRL_LIB=$L KIND=ant VARIANTS=auto,auto,auto python tools/train_only.py >> gpurun_out/r05_train_ant1.log 2>&1
done; done
tail -3 gpurun_out/r05_t2.log; grep "==\|us/step\|cycles" gpurun_out/r05_train_ant1.log
