import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from allset_amd import dense
dev = torch.device("cuda:0")
torch.manual_seed(0)
n, K, N = 3001, 128, 128
x = torch.randn(n, K, device=dev); W = torch.randn(N, K, device=dev) / 11; b = torch.zeros(N, device=dev)
g = 1 + 0.2 * torch.randn(K, device=dev); bt = 0.3 * torch.randn(K, device=dev); G = torch.randn(n, N, device=dev)
res = {}
for mode in ("f32", "bf16x6"):
    os.environ["ALLSET_DENSE_MFMA"] = mode
    for rep in range(3):
        y, st = dense.fused_linear_fwd(x, W, b, g, bt)
        gx, dg, db = dense.fused_linear_bwd(G, None, 0.0, W, x, st, g, False, 0.0, 0)
        torch.cuda.synchronize()
        res[(mode, rep)] = (y.clone(), st.clone(), gx.clone(), dg.clone(), db.clone())
ry, rst, rgx, rdg, rdb = res[("f32", 0)]
for rep in range(3):
    y, st, gx, dg, db = res[("bf16x6", rep)]
    bad_st = ((st - rst).abs() > 1e-4).any(1).nonzero().flatten().tolist()
    bad_gx = ((gx - rgx).abs() > 1e-3).any(1).nonzero().flatten().tolist()
    print(rep, "y err", float((y - ry).abs().max()), "bad stats rows", bad_st[:10], "bad gx rows", bad_gx[:12], [r % 16 for r in bad_gx[:12]],
          "dg err", float((dg - rdg).abs().max()), "db err", float((db - rdb).abs().max()))
    # bwd with the reference stats
    gx2, _, _ = dense.fused_linear_bwd(G, None, 0.0, W, x, rst, g, False, 0.0, 0)
    print("   with f32-path stats: bad gx rows", ((gx2 - rgx).abs() > 1e-3).any(1).nonzero().flatten().tolist()[:12])
print("---- detail")
os.environ["ALLSET_DENSE_MFMA"] = "bf16x6"
y, st = dense.fused_linear_fwd(x, W, b, g, bt)
torch.set_printoptions(linewidth=250, precision=3, sci_mode=False)
for trial in range(3):
    gx, dg, db = dense.fused_linear_bwd(G, None, 0.0, W, x, st, g, False, 0.0, 0)
    d = (gx - rgx).abs()
    bad = (d > 1e-3).any(1).nonzero().flatten().tolist()
    for r in bad[:2]:
        big = (d[r] > 0.3 * d[r].max()).nonzero().flatten().tolist()
        print(trial, "row", r, "max", float(d[r].max()), "median", float(d[r].median()), "cols with big error", big[:16])
    cd = (dg - rdg).abs()
    print("   dg bad cols", (cd > 1e-2).nonzero().flatten().tolist()[:24], " db bad cols", ((db - rdb).abs() > 1e-2).nonzero().flatten().tolist()[:10])
