cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in base hnoslp hlate; do
  if [ $v = base ]; then L=""; else L=icrl_amd/lib/var/libicrl_$v.so; fi
  echo -n "$v: "; ICRL_LIB=$L VARIANTS=auto,auto,auto python tools/train_only.py 2>&1 | grep "us/step" | awk '{printf "%s ", $4}'; echo
done; done
timeout 1200 python -m pytest tests/test_seed_batch_gpu.py tests/test_ppo_train_gpu.py tests/test_fullsize_parity_gpu.py -x -q -m gpu 2>&1 | tail -6
timeout 600 python bench.py --no_cpu_baseline --no_configs2 --no_configs3 --no_configs4 --no_generic > gpurun_out/r05_bench_b.json 2>/dev/null
python -c "
import json; j=json.load(open('gpurun_out/r05_bench_b.json')); print(j['value'], j['ms_per_step'], j['roofline_ppo']['us_per_optimizer_step'], j['early_stop_fraction'], j['value_no_early_stop'], j['seed_batch']['aggregate_env_steps_per_s'])"
