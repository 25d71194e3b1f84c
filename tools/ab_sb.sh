export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; rm -rf gpurun_out/pmc_lds*
export E4S_SB_SWP=0
bash tools/pmc_pass.sh "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" pmc_lds1
bash tools/pmc_pass.sh "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" pmc_lds2
cd $R; python tools/rocpd_pmc.py gpurun_out/pmc_lds1/pmc_results.db 2>&1 | grep -A7 "4, 1, 1, 8, 5\|<1, 2, 1, 4, 5, 2, true, false" | head -40
python tools/rocpd_pmc.py gpurun_out/pmc_lds2/pmc_results.db 2>&1 | grep -A7 "4, 1, 1, 8, 5\|<1, 2, 1, 4, 5, 2, true, false" | head -40
rm -rf gpurun_out/pmc_lds1 gpurun_out/pmc_lds2
