cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/r05_bench_c.json 2> gpurun_out/r05_bench_c.err
timeout 1400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r05_full3.log
cat gpurun_out/r05_full3.log; head -c 600 gpurun_out/r05_bench_c.json
