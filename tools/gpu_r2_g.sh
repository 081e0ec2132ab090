R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python3 -m pytest tests -m gpu -x -q > $O/r2g_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/r2g_tests.log
python3 tools/fp32_probe.py 5 1024,4096,16384 > $O/r2g_fp32_c5.log 2>&1; cat $O/r2g_fp32_c5.log
python3 tools/fp32_probe.py 3 4096,16384 > $O/r2g_fp32_c3.log 2>&1; cat $O/r2g_fp32_c3.log
