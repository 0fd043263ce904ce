import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from types import SimpleNamespace
import cases
from allset_amd import SetGNN
dev = torch.device("cuda:0")
for name in ("cora_ds_add", "citeseer_pma_h4"):
    case = cases.build_case(name)
    model = SetGNN(case["args"]).to(dev).eval()
    data = SimpleNamespace(x=torch.from_numpy(case["x"]).to(dev), edge_index=torch.from_numpy(case["edge_index"]).to(dev),
                           norm=torch.from_numpy(case["norm"]).to(dev))
    with torch.no_grad():
        ref = model(data)                       # warm-up: builds the incidence, caches weights
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): model(data)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = model(data)
        g.replay(); torch.cuda.synchronize()
        print(name, "graph == eager:", torch.equal(out, ref), float((out - ref).abs().max()))
        t0 = time.perf_counter()
        for _ in range(200): g.replay()
        torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 200
        t0 = time.perf_counter()
        for _ in range(200): model(data)
        torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 200
        print(f"  eager {te*1e3:.3f} ms   graph replay {tg*1e3:.3f} ms")
