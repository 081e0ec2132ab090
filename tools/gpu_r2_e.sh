R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
for q in 3 4 5; do
  for hq in 4 8; do
  echo "== GOGP_EXP_PRIO=$q GPU_MAX_HW_QUEUES=$hq"
  GPU_MAX_HW_QUEUES=$hq GOGP_EXP_PRIO=$q python3 tools/batch_probe.py 2 4096 2,4,6,8 2>&1 | grep "k=[0-9]*:" 
  done
done
