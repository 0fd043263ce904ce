set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_dist_random.py -x -q > $OUT/r04c_pytest.txt 2>&1; tail -5 $OUT/r04c_pytest.txt
for c in 1 2 4; do
  SIM_WORLDS=8 SIM_T1_MS=10.04 SIM_CHUNKS=$c SIM_KERNELS=1 timeout 600 python tools/sim_rank.py deepsets columns 2>&1 | grep -v amdgpu.ids
done > $OUT/r04_sim_rank_chunks.txt
cat $OUT/r04_sim_rank_chunks.txt
echo finished
