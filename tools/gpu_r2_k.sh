R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python3 -m pytest tests/test_sharded.py -m gpu -x -q > $O/r2k_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r2k_tests.log
python3 tools/sharded_probe.py 3 2>&1 | grep sharded
python3 tools/sharded_probe.py 4 2>&1 | grep sharded
