export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_se -o se -- python3 $R/tools/probes/time_se_gate.py > $R/gpurun_out/prof_se.log 2>&1
cd $R
python tools/rocpd_by_grid.py gpurun_out/prof_se/se_results.db 0.0 | cut -c1-200 > gpurun_out/se_gate_dur.txt
rm -rf gpurun_out/prof_se
cat gpurun_out/se_gate_dur.txt | head -12
