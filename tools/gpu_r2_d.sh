set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch" 2>&1 | tail -3
python3 tools/batch_probe.py 2 > $O/r2d_batch_c2.log 2>&1; cat $O/r2d_batch_c2.log
python3 tools/batch_probe.py 2 8192 > $O/r2d_batch_8192.log 2>&1; cat $O/r2d_batch_8192.log
python3 bench.py --config 1 --no-cpu-baseline | cut -c1-400
