#!/bin/bash
# Round-end records on the GPU box, one gpurun call:  gpurun --timeout 3600 -- 'bash tools/round_end.sh <tag> [suite]'
#   the full -m gpu suite (when "suite" is given) + smoke, the default bench line, and every rocprofv3 record the line's
#   rooflines are recomputed from (tools/collect_profiles.sh).  Outputs under gpurun_out/; copy what is kept into profiles/.
TAG=${1:-r06}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
if [[ "${2:-}" == suite ]]; then
  python -m pytest tests -q -m gpu --durations=8 > gpurun_out/${TAG}_gpu_suite.log 2>&1; tail -14 gpurun_out/${TAG}_gpu_suite.log
  python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/${TAG}_gpu_suite.log 2>&1; tail -1 gpurun_out/${TAG}_gpu_suite.log
fi
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo bench rc=$?
bash tools/collect_profiles.sh ${TAG} C3 > gpurun_out/${TAG}_collect.log 2>&1; tail -3 gpurun_out/${TAG}_collect.log
