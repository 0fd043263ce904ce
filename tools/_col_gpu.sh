python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py -x -q -m gpu -k "block_transpose or colsharded" 2>&1 | tail -3
python tools/sim_rank.py deepsets columns 2>&1 | grep -v Warn | tail -2
python tools/sim_rank.py pma columns 2>&1 | grep -v Warn | tail -2
