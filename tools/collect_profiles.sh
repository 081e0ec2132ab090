# Copies what tools/profile_round3.sh left under gpurun_out/prof_r3 into profiles/ under the tracked names
# (profiles/README.md, round 3).  usage: bash tools/collect_profiles.sh
set -e
O=gpurun_out/prof_r3; P=profiles
newest() { ls -t $1 | head -1; }
for c in 1 2 3 4 5; do tail -n 1 $O/bench_c$c.json > $P/r03_bench_c$c.json; done
cp "$(newest "$O/stats_c3/*/*_kernel_stats.csv")" $P/r03_c3_kernel_stats.csv
cp "$(newest "$O/stats_c3/*/*_kernel_trace.csv")" $P/r03_c3_kernel_trace.csv
cp "$(newest "$O/stats_c2/*/*_kernel_stats.csv")" $P/r03_c2_kernel_stats.csv
cp "$(newest "$O/stats_c2k8/*/*_kernel_stats.csv")" $P/r03_c2_candidates8_kernel_stats.csv
cp "$(newest "$O/stats_c5/*/*_kernel_stats.csv")" $P/r03_c5_fp32_n32768_kernel_stats.csv
cp $O/pmc_summary_c3.txt $P/r03_pmc_summary_c3.txt
cp $O/pmc_summary_c5.txt $P/r03_pmc_summary_c5_fp32_n16384.txt
cp $O/roofline_c3.json $P/r03_c3_roofline.json
cp $O/roofline_c2.json $P/r03_c2_roofline.json
cp $O/roofline_c5_n32768.json $P/r03_c5_fp32_n32768_roofline.json
cp $O/pmc_traffic.json $P/r03_pmc_traffic.json
cp $O/gemm_bench.txt $P/r03_gemm_bench.txt
ls -la $P | grep r03_
