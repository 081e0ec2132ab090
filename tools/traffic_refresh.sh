set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FAST="--no-cpu-baseline --no-produce --candidates 1"
rm -rf $O/pmc_fetch_c3 $O/pmc_write_c3 $O/pmc_fetch_c5 $O/pmc_write_c5
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_c3 -- python3 $R/bench.py --steps 2 --warmup 1 $FAST > $O/pmc_fetch_c3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_write_c3 -- python3 $R/bench.py --steps 2 --warmup 1 $FAST > $O/pmc_write_c3.log 2>&1; echo "pmc traffic c3 done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_c5 -- python3 $R/bench.py --config 5 --nobs 16384 --steps 2 --warmup 1 $FAST > $O/pmc_fetch_c5.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_write_c5 -- python3 $R/bench.py --config 5 --nobs 16384 --steps 2 --warmup 1 $FAST > $O/pmc_write_c5.log 2>&1; echo "pmc c5 done"
cd $R
python3 tools/pmc_traffic.py 3 $O/pmc_fetch_c3 $O/pmc_write_c3 dgemm_nt_kernel 16384 > $O/t3.json
python3 tools/pmc_traffic.py 5_at_N16384 $O/pmc_fetch_c5 $O/pmc_write_c5 sgemm_nt_kernel 16384 > $O/t5.json
python3 - <<PY
import json, sys
sys.path.insert(0, "$R")
from gogp_amd import _lib
v = _lib.lib().gogp_version().decode()
d = {"build": v.split("build ")[-1], "library": v}
d.update(json.load(open("$O/t3.json"))); d.update(json.load(open("$O/t5.json")))
json.dump(d, open("$O/pmc_traffic.json", "w"), indent=1)
json.dump(d, open("$R/profiles/r04_pmc_traffic.json", "w"), indent=1)
print(v)
PY
python3 bench.py > $O/bench_c3_final.json 2> $O/bench_c3_final.err
find $O -name "*counter_collection.csv" -size +6M -delete
find $O -name "*kernel_trace.csv" -size +6M -delete
find $O -name "*.db" -delete
tail -c 300 $O/bench_c3_final.json
