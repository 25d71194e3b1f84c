# kernel table of the encoder alone (tools/time_encoder.py) -> gpurun_out/${1}_encoder_kernel_stats.txt
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp
export PYTHONPATH=$R
python3 $R/tools/time_encoder.py > $R/gpurun_out/${1:-r04}_encoder_timing.txt 2>&1
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_enc -o enc -- python3 $R/tools/time_encoder.py > $R/gpurun_out/prof_enc.log 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/prof_enc/enc_results.db | cut -c1-240 > gpurun_out/${1:-r04}_encoder_kernel_stats.txt
rm -rf gpurun_out/prof_enc
cat gpurun_out/${1:-r04}_encoder_timing.txt | tail -2
