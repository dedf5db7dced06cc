#!/bin/bash
# A/B of the library against the previous build kept at ringsnark_amd/librs_hip_prev.so: witness-map parity tests, then the headline probe with each
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ab
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_witness_large.py tests/test_incomplete.py tests/test_scalar_wires.py tests/test_config_scale.py -q -m gpu -x > gpurun_out/ab/tests.log 2>&1; tail -3 gpurun_out/ab/tests.log
python tools/headline_probe.py 16 13 3 C3 > gpurun_out/ab/new.txt 2>&1
RINGSNARK_AMD_LIB=ringsnark_amd/librs_hip_prev.so python tools/headline_probe.py 16 13 3 C3 > gpurun_out/ab/prev.txt 2>&1
python tools/headline_probe.py 16 13 3 C3 > gpurun_out/ab/new2.txt 2>&1
for f in prev new new2; do echo == $f; grep -E "^proof|io_mid_out|r1cs_eval_cols|transpose_out" gpurun_out/ab/$f.txt; done
