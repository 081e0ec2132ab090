# Copies what tools/profile_round3.sh left under gpurun_out/prof_r${R:-4} into profiles/ under the tracked names
# (profiles/README.md, round ${R:-4}).  usage: bash tools/collect_profiles.sh
set -e
O=gpurun_out/prof_r${R:-4}; P=profiles
newest() { ls -t $1 | head -1; }
for c in 1 2 3 4 5; do tail -n 1 $O/bench_c$c.json > $P/r0${R:-4}_bench_c$c.json; done
cp "$(newest "$O/stats_c3/*/*_kernel_stats.csv")" $P/r0${R:-4}_c3_kernel_stats.csv
cp "$(newest "$O/stats_c3/*/*_kernel_trace.csv")" $P/r0${R:-4}_c3_kernel_trace.csv
cp "$(newest "$O/stats_c2/*/*_kernel_stats.csv")" $P/r0${R:-4}_c2_kernel_stats.csv
cp "$(newest "$O/stats_c2k8/*/*_kernel_stats.csv")" $P/r0${R:-4}_c2_candidates8_kernel_stats.csv
cp "$(newest "$O/stats_c5/*/*_kernel_stats.csv")" $P/r0${R:-4}_c5_fp32_n32768_kernel_stats.csv
cp $O/pmc_summary_c3.txt $P/r0${R:-4}_pmc_summary_c3.txt
if [ "${R:-4}" -ge 5 ]; then cp $O/pmc_summary_c5.txt $P/r0${R:-4}_pmc_summary_c5.txt; else cp $O/pmc_summary_c5.txt $P/r0${R:-4}_pmc_summary_c5_fp32_n16384.txt; fi  # round 5: config 5's own N = 65536
cp $O/roofline_c3.json $P/r0${R:-4}_c3_roofline.json
cp $O/roofline_c2.json $P/r0${R:-4}_c2_roofline.json
cp $O/roofline_c5_n32768.json $P/r0${R:-4}_c5_fp32_n32768_roofline.json
cp $O/pmc_traffic.json $P/r0${R:-4}_pmc_traffic.json
[ -f $O/gemm_bench.txt ] && cp $O/gemm_bench.txt $P/r0${R:-4}_gemm_bench.txt
for f in gemm_fixedcost graph_probe produce_probe mixed_probe produce_small_probe split_probe panel_probe valu_cost fp32_bias_probe; do [ -f $O/$f.txt ] && cp $O/$f.txt $P/r0${R:-4}_$f.txt; done
ls -la $P | grep r0${R:-4}_
