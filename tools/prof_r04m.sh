cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
for L in 0 0.9375 0.96875; do
  SIM_WORLDS=8 SIM_T1_MS=11.95 SIM_LOCALITY=$L timeout 600 python tools/sim_rank.py pma rows 2>&1 | grep -v amdgpu.ids
done > $OUT/r04_sim_rank_halo_pma.txt
cat $OUT/r04_sim_rank_halo_pma.txt
