# FETCH_SIZE calibration on known byte counts (tools/probes/fetch_calib_probe.hip) -> gpurun_out/fetch_calib.txt
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
[ -x $R/tools/probes/fetch_calib_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $R/tools/probes/fetch_calib_probe.hip -o $R/tools/probes/fetch_calib_probe
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/calib -o calib -- $R/tools/probes/fetch_calib_probe > $R/gpurun_out/calib.log 2>&1
cd $R
python tools/rocpd_pmc.py gpurun_out/calib/calib_results.db > gpurun_out/fetch_calib_raw.txt
python - <<'PY' > gpurun_out/fetch_calib.txt
import re
true_kb = (1 << 30) / 1024
cur = None
print("FETCH_SIZE reported / true bytes, per access pattern (1 GiB read once per launch; tools/probes/fetch_calib_probe.hip)")
for line in open("gpurun_out/fetch_calib_raw.txt"):
    if not line.startswith(" "):
        cur = line.strip()
    else:
        m = re.match(r"\s+FETCH_SIZE\s+calls=\s*(\d+)\s+avg=\s*([\d.]+)", line)
        if m:
            print(f"{float(m.group(2)) / true_kb:6.3f}   {cur}")
PY
rm -rf gpurun_out/calib
cat gpurun_out/fetch_calib.txt
