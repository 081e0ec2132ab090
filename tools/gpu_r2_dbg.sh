cd $GRAFT_REPO_ROOT
for t in "test_rccl_transport_on_one_rank" "test_sharded_grid_matches_single_gpu[1x1]" "test_sharded_grid_matches_single_gpu[2x2]" "test_sharded_not_positive_definite_reaches_every_rank" "test_sharded_over_gloo_processes[2]"; do
  python3 -X faulthandler -m pytest "tests/test_sharded.py::$t" -m gpu -x -q > gpurun_out/dbg_$$.log 2>&1
  echo "== $t rc=$?"; tail -4 gpurun_out/dbg_$$.log
done
python3 -c "
import sys; sys.path.insert(0,'.')
from gogp_amd import _lib
L=_lib.lib(); print('loaded only')"
echo "== load-only rc=$?"
python3 -c "
import sys; sys.path.insert(0,'.')
import numpy as np
from gogp_amd import gp, kernel
g=gp.GP(1, kernel.Normal, kernel.ConstantNoise(0.1), ThetaSimil=[1.0]); g.Absorb(np.zeros((2,1))+[[0],[1]], [0.,1.]); print(g.LML()); g.close()"
echo "== plain GP rc=$?"
exit 0
