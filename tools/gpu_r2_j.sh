R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python3 -m pytest tests -m gpu -x -q > $O/r2j_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r2j_tests.log
for c in 2 3; do
python3 bench.py --config $c --no-cpu-baseline --no-produce --candidates 1 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('config', d['config']['baseline_config'], 'evals/s %.3f ms %.3f frac %.4f frac_wall %.4f busy %.3f sum %.3f launches %.0f' % (d['value'], d['ms_per_step'], r['frac'], r['frac_wall'], r['kernel_busy_ms_per_step'], r['sum_of_launch_durations_ms_per_step'], r['launches_per_step']))"
done
