cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r05 > gpurun_out/profile_r05.log 2>&1
timeout 900 python bench.py > gpurun_out/r05_bench_d.json 2> gpurun_out/r05_bench_d.err
timeout 1400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r05_full4.log
cat gpurun_out/r05_full4.log; head -c 300 gpurun_out/r05_bench_d.json
