cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ppo_train_gpu.py tests/test_gae_gpu.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r05_t1.log
timeout 600 python bench.py > gpurun_out/r05_bench_a.json 2> gpurun_out/r05_bench_a.err
python tools/train_only.py > gpurun_out/r05_train_hc0.log 2>&1
KIND=ant python tools/train_only.py > gpurun_out/r05_train_ant0.log 2>&1
tail -3 gpurun_out/r05_t1.log; head -c 1500 gpurun_out/r05_bench_a.json; grep us/step gpurun_out/r05_train_hc0.log gpurun_out/r05_train_ant0.log
