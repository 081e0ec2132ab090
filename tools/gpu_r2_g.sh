R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fp32" > $O/r2g_tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/r2g_tests.log
(python3 tools/fp32_probe.py 5 4096,16384 1 break; python3 tools/fp32_probe.py 3 4096,16384 1 break) 2>&1 | grep -v amdgpu.ids > $O/r2g_fp32_break.log; cat $O/r2g_fp32_break.log
