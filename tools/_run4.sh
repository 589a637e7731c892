cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ppo_train_gpu.py tests/test_lap_grid_gpu.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r05_t5.log
tail -12 gpurun_out/r05_t5.log
VARIANTS=auto,pairs,auto,pairs python tools/train_only.py 2>&1 | grep "us/step\|cycles" | tee gpurun_out/r05_train_hc1.log
