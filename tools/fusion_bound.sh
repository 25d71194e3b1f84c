# Upper bound of what fusing `512 -> 1024 up` + `32 -> 32 @1024^2 conv` into one kernel could save: the tuning library (-DE4S_PHASE_PROF) with the conv's activation DMA
# switched off (E4S_CHAIN_EXP=4: everything the fusion would remove from it) and with the up kernel's output stores switched off (E4S_HC_EXP=4), batch 4, in isolation.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
export E4S_HIP_LIB=$R/e4s2024_amd/lib/libe4s_hip_prof.so
if [ "${1:-all}" != "hc" ]; then
for e in 0 4 2 1 7; do echo "E4S_CHAIN_EXP=$e (1 = no epilogue, 2 = no MFMAs, 4 = no activation DMA)"; E4S_CHAIN_EXP=$e timeout 120 python tools/time_chain.py 2>&1 | grep "chain" | grep "fused rgb" | head -2 | cut -c1-160; done
fi
for e in 0 4 2 1 7; do echo "E4S_HC_EXP=$e (1 = no epilogue, 2 = no MFMAs, 4 = no output stores)"; E4S_HC_EXP=$e timeout 120 python tools/time_chain.py 2>&1 | grep "old fused up" | head -2 | cut -c1-160; done
