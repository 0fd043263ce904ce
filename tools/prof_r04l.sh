cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
for L in 0 0.9375 0.96875; do
  SIM_WORLDS=8 SIM_T1_MS=10.04 SIM_LOCALITY=$L SIM_KERNELS=1 timeout 600 python tools/sim_rank.py deepsets rows 2>&1 | grep -v amdgpu.ids
done > $OUT/r04_sim_rank_halo.txt
SIM_WORLDS=2,4 SIM_T1_MS=10.04 SIM_LOCALITY=0.9375 timeout 600 python tools/sim_rank.py deepsets rows 2>&1 | grep -v amdgpu.ids >> $OUT/r04_sim_rank_halo.txt
cat $OUT/r04_sim_rank_halo.txt
