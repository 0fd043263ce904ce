"""GPU: the hyper-parameter combinations the reference's own experiment scripts train with -- every dataset branch of
``/root/reference/src/run_AllSetTransformer.sh:21-300`` (heads in {1, 4, 8}, ``MLP_hidden`` in {64, 128, 256, 512},
``Classifier_hidden`` in {64, 128, 256}, one layer, 2-layer MLPs) and the same widths for AllDeepSets (``run_all_experiments.sh``
sweeps the method over the same hidden sizes) -- on a dataset-shaped hypergraph with self-loop hyperedges, product (HIP kernels
through the C ABI) against the live oracle: logits, input gradient and every parameter gradient at 1e-4.  The fixtures pin a few
of these widths; this pins the matrix the reference actually runs (which dense kernel family a width takes differs: fused
64 / 128, tiled 256 / 512; heads 8 at 128 puts 16 columns in a head)."""
import zlib

import numpy as np
import pytest
import torch

import cases
import util

pytestmark = pytest.mark.gpu
RTOL = ATOL = 1e-4

# (dataset branch of the script, heads, MLP_hidden, Classifier_hidden)
SCRIPT_CONFIGS = [
    ("cora", 4, 256, 128), ("citeseer", 8, 512, 256), ("pubmed", 8, 256, 256), ("coauthor_cora", 8, 128, 128),
    ("coauthor_dblp", 8, 512, 128), ("zoo", 1, 64, 64), ("20newsW100", 8, 256, 64), ("Mushroom", 1, 128, 128),
    ("NTU2012", 1, 256, 128), ("ModelNet40", 8, 512, 128), ("yelp", 1, 64, 64), ("house-committees-100", 8, 512, 256),
    ("walmart-trips-100", 8, 256, 128),
]


def _case(tag, mode, heads, hidden, chid, attempt=0):
    name = f"script_{tag}_{mode}" + (f"_{attempt}" if attempt else "")
    seed = zlib.crc32(name.encode()) & 0x7FFFFFFF
    rng = np.random.default_rng(seed)
    n_v, f, k = 260, 40, 5
    ei = cases.random_hypergraph(rng, n_v, 90, 1100, True)           # + one self-loop hyperedge per vertex (train.py's default)
    x = rng.standard_normal((n_v, f)).astype(np.float32)
    args = cases.make_args(mode, f, hidden, k, heads=heads, Classifier_hidden=chid)
    # kinkfree: cases.kinkfree_biases -- what this matrix pins is which kernel family each width / head count takes, not relu
    # patterns (the fixtures' job); with the pre-activations away from zero the first draw is stable (the re-draw loop below used
    # to evaluate the float64 oracle up to 90 times on the 512-wide branches: 20 s of a 1200 s budget for one test)
    return dict(name=name, args=args, x=x, edge_index=ei, norm=np.ones(ei.shape[1], dtype=np.int64), seed=seed, big=False, kinkfree=True)


@pytest.mark.parametrize("mode", ["pma", "ds_add"])
@pytest.mark.parametrize("tag,heads,hidden,chid", SCRIPT_CONFIGS)
def test_run_script_configuration_matches_the_oracle(tag, heads, hidden, chid, mode, device):
    from allset_amd import SetGNN
    from oracle import allset_oracle as oracle
    if mode == "ds_add":
        heads = 1
    # Parity is only meaningful where the model is smooth: a relu input within fp32 rounding of zero (there are ~10^5 relu
    # inputs per example) makes the exact gradient itself jump -- the first data drawn for the citeseer branch has one: its
    # FLOAT64 gradient moves by 1.3 % of its scale under a 1e-7 perturbation of x, in one direction only.  Such examples are
    # re-drawn (next seed), as tests/test_gpu_random_shapes.py rejects them -- accepted or refused by util.oracle_is_smooth_here (the
    # float64 oracle alone, before the product has run), then compared ONCE.
    for attempt in range(30):
        case = _case(tag, "pma_h1" if mode == "pma" else "ds_add", heads, hidden, chid, attempt)
        case["args"].heads = heads
        spec = [(k, tuple(v.shape)) for k, v in SetGNN(case["args"]).state_dict().items()]
        sd = {k: torch.from_numpy(v) for k, v in cases.make_state_dict(spec, case["seed"], case.get("kinkfree", False)).items()}
        o64 = util.run_oracle(case, sd, torch.float64)
        G = torch.from_numpy(cases.cotangent(case["name"], o64["logits"].shape))
        if util.oracle_is_smooth_here(sd, case["args"], case["x"], case["edge_index"], case["norm"], G, seed=case["seed"]):
            break
    else:
        pytest.fail("no kink-free example in thirty draws: the generator of this test is broken, not the product")
    res = util.run_product(case, sd, device)
    orc = util.run_oracle(case, sd)
    # Yardstick: the oracle in float64.  Allowed distance: 1e-4 of the tensor's scale, or -- where the fp32 ORACLE itself is
    # further than that from float64 (512-wide LayerNorm stacks: rounding is amplified by rstd, tests/test_gpu_random_shapes.py)
    # -- three times the fp32 oracle's own distance: fp32 parity means "as close to the exact result as the reference's fp32".
    worst = {}

    def check(got, e32, e64, what, floor):
        scale = max(float(e64.abs().max()), floor, 1e-3)
        own = float((e32.double() - e64).abs().max())
        dist = float((got.double() - e64).abs().max())
        worst[what] = (dist / scale, own / scale)
        assert dist <= max(ATOL * scale, 3.0 * own), f"{case['name']}/{what}: product {dist:.3e} from float64, fp32 oracle {own:.3e}, scale {scale:.3e}"

    for key in ("logits", "grad_x"):
        check(res[key], orc[key], o64[key].detach(), key, 0.0)
    gscale = max(float(v.abs().max()) for v in o64["grads"].values())
    for key, gexp in o64["grads"].items():
        if key in res["grads"]:
            check(res["grads"][key], orc["grads"][key].detach(), gexp.detach(), "grad " + key, 1e-2 * gscale)
    # and the product is not systematically worse than the reference's own fp32 arithmetic
    dp, do = max(v[0] for v in worst.values()), max(v[1] for v in worst.values())
    assert dp <= max(2e-4, 3.0 * do), (dp, do)
