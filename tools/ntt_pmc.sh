cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for P in C4 C3; do
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05_ntt_pmc_$P/stall -o run --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -- python3 tools/ntt16_pmc.py $P 1 3 > gpurun_out/r05_ntt_pmc_$P.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05_ntt_pmc_$P/mix -o run --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM -- python3 tools/ntt16_pmc.py $P 1 3 >> gpurun_out/r05_ntt_pmc_$P.log 2>&1
done
find gpurun_out/r05_ntt_pmc_* -name '*.db' -delete
ls -R gpurun_out/r05_ntt_pmc_C4 | head
