# HIP API calls of the parser alone (names and counts) -> gpurun_out/api_trace.txt
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp
rocprofv3 --hip-runtime-trace --kernel-trace -d $R/gpurun_out/prof_api -o api -- python3 $R/tools/time_parts.py parse 16 10 > $R/gpurun_out/prof_api.log 2>&1
cd $R
python - <<'PY' > gpurun_out/api_trace.txt
import sqlite3, collections
con = sqlite3.connect("gpurun_out/prof_api/api_results.db")
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if not t.startswith("rocpd_") or t.count("_") < 3][:40])
for t in ("regions", "regions_and_samples", "top", "hip_api", "rocpd_region"):
    if t in tabs:
        cols = [d[1] for d in con.execute(f"pragma table_info('{t}')")]
        print(t, cols)
        if "name" in cols:
            for n, c in con.execute(f"select name, count(*) from {t} group by name order by 2 desc limit 30"):
                print(c, n)
        break
PY
rm -rf gpurun_out/prof_api
head -50 gpurun_out/api_trace.txt
