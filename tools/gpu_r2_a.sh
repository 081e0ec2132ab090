set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python3 -m pytest tests -m gpu -x -q > $O/r2a_tests.log 2>&1 || { tail -30 $O/r2a_tests.log; exit 1; }
tail -3 $O/r2a_tests.log
python3 bench.py --config 2 > $O/r2a_bench_c2.json 2> $O/r2a_bench_c2.err || { tail -20 $O/r2a_bench_c2.err; exit 1; }
cat $O/r2a_bench_c2.json
python3 bench.py > $O/r2a_bench_c3.json 2> $O/r2a_bench_c3.err || { tail -20 $O/r2a_bench_c3.err; exit 1; }
cat $O/r2a_bench_c3.json
