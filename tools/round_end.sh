cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r05_gpu_suite.log 2>&1; tail -14 gpurun_out/r05_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/r05_gpu_suite.log 2>&1; tail -1 gpurun_out/r05_gpu_suite.log
python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err; echo bench rc=$?
bash tools/collect_profiles.sh r05 C3 > gpurun_out/r05_collect.log 2>&1; tail -3 gpurun_out/r05_collect.log
