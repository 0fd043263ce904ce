import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_train_parity as T
from allset_amd import dense
seeds = []
real = dense._draw_seed
dense._draw_seed = lambda: (seeds.append(real()) or seeds[-1])
dev = torch.device("cuda:0")
for over in (dict(MLP_num_layers=3, MLP_hidden=128), dict(MLP_num_layers=1), dict(All_num_layers=2)):
    for leaf in (False, True):
        for attempt in range(3):
            seeds.clear()
            try:
                T._one_training_step("cora_ds_add", over, dev, seeds, attempt, False, leaf_x=leaf)
                print(over, "leaf" if leaf else "grad", attempt, "OK")
            except AssertionError as e:
                msg = str(e).split("\n")
                print(over, "leaf" if leaf else "grad", attempt, "FAIL", msg[0][:80], [m for m in msg if "Mismatched" in m or "absolute" in m])
