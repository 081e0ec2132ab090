R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python3 bench.py --config 5 > $O/r2h_bench_c5.json 2> $O/r2h_bench_c5.err; echo "c5 rc=$?"; cut -c1-1200 $O/r2h_bench_c5.json; tail -3 $O/r2h_bench_c5.err
python3 bench.py --config 4 > $O/r2h_bench_c4.json 2> $O/r2h_bench_c4.err; echo "c4 rc=$?"; cut -c1-1200 $O/r2h_bench_c4.json; tail -3 $O/r2h_bench_c4.err
python3 bench.py --config 2 > $O/r2h_bench_c2.json 2> $O/r2h_bench_c2.err; echo "c2 rc=$?"; python3 -c "
import json; d=json.load(open('$O/r2h_bench_c2.json')); print(d['value'], d.get('concurrent_candidates'), d['cpu_baseline']['seconds'])"
python3 bench.py > $O/r2h_bench_c3.json 2> $O/r2h_bench_c3.err; echo "c3 rc=$?"; python3 -c "
import json; d=json.load(open('$O/r2h_bench_c3.json')); print(d['value'], d['roofline']['frac'], d['roofline']['frac_wall'], d['cpu_baseline'], d['parity_vs_oracle'])"
