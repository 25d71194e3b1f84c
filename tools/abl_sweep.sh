# every tuning library under e4s2024_amd/lib/libe4s_abl*.so (tools/build_abl.sh) through tools/time_mx_abl.py, the product library first and last
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
python tools/time_mx_abl.py 2>&1 | tail -1
for f in $(ls e4s2024_amd/lib/libe4s_abl*.so | sort -V); do E4S_HIP_LIB=$f python tools/time_mx_abl.py 2>&1 | tail -1; done
python tools/time_mx_abl.py 2>&1 | tail -1
