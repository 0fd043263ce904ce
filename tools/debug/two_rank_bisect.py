"""Debug: run subsets of tests/test_gpu_two_ranks.py's configuration lists in one two-rank spawn and apply its own model check."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

def worker(rank, world, port, q, layer_idx, model_idx):
    import test_gpu_two_ranks as T
    T.LAYER_CFGS = [T.LAYER_CFGS[i] for i in layer_idx]
    T.MODEL_CFGS = [T.MODEL_CFGS[i] for i in model_idx]
    T._worker(rank, world, port, q)

if __name__ == "__main__":
    import torch.multiprocessing as mp
    import test_gpu_two_ranks as T
    nL, nM = len(T.LAYER_CFGS), len(T.MODEL_CFGS)
    trials = {"bn-cols alone": ([], [nM - 1]), "all layers + bn-cols": (list(range(nL)), [nM - 1]), "all models": ([], list(range(nM))),
              "models 0..5 w/o rows-bn": ([], list(range(nM - 2)) + [nM - 1])}
    for arg in sys.argv[1:]:
        name, l, m = arg.split("/")
        trials = {name: ([int(v) for v in l.split(",") if v], [int(v) for v in m.split(",") if v])}
    for name, (li, mi) in trials.items():
        ctx = mp.get_context("spawn"); q = ctx.Queue(); port = T._free_port()
        ps = [ctx.Process(target=worker, args=(r, 2, port, q, li, mi)) for r in range(2)]
        [p.start() for p in ps]
        got = {}
        for _ in range(2):
            rank, res = q.get(timeout=400)
            if isinstance(res, str): print(res); sys.exit(1)
            got[rank] = res
        [p.join(60) for p in ps]
        for i in mi:
            cfg = T.MODEL_CFGS[i]
            try:
                T.test_two_rank_sharded_setgnn_equals_oracle(cfg, got)
                print(f"[{name}] {cfg[:3]} {sorted(cfg[3])}: ok", flush=True)
            except AssertionError as e:
                print(f"[{name}] {cfg[:3]} {sorted(cfg[3])}: FAIL {str(e).splitlines()[0][:120]}", flush=True)
