# one PMC pass over the regional-style encoder alone (tools/time_encoder.py, 16 faces): counters + kernel trace only
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp
rocprofv3 --kernel-trace --pmc $1 -d $R/gpurun_out/$2 -o pmc -- python3 $R/tools/time_encoder.py > $R/gpurun_out/$2.log 2>&1
cd $R; python tools/rocpd_pmc.py gpurun_out/$2/pmc_results.db conv3x3_mx3 > gpurun_out/$2.txt; rm -rf gpurun_out/$2; cat gpurun_out/$2.txt
