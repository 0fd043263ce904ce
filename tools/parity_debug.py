"""Per-tensor parity report of one tests/cases.py case on the GPU (eval mode): product vs live oracle, every gradient.
Usage: python tools/parity_debug.py mid4k_ds_add [...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
import util  # noqa: E402

dev = torch.device("cuda:0")
if os.environ.get("ALLSET_ARITH"):
    from allset_amd import dense
    dense.set_arithmetic(os.environ["ALLSET_ARITH"])
    print("arithmetic:", dense.get_arithmetic())
for name in sys.argv[1:]:
    case, g = cases.build_case(name), util.load_golden(name)
    sd = util.state_dict_for(case, g)
    res = util.run_product(case, sd, dev)
    orc = util.run_oracle(case, sd)
    print(f"== {name}")
    for k in ("logits", "v2e0", "e2v0", "grad_x"):
        d = float((res[k] - orc[k]).abs().max())
        print(f"  {k:50s} max|diff| {d:.3e}  scale {float(orc[k].abs().max()):.3e}")
    for k, ge in orc["grads"].items():
        if k in res["grads"]:
            d = float((res["grads"][k] - ge.detach()).abs().max())
            print(f"  grad {k:45s} max|diff| {d:.3e}  scale {float(ge.abs().max()):.3e}  rel {d / max(float(ge.abs().max()), 1e-30):.2e}")
