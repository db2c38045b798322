"""fused row-panel MLP forward vs the layer-by-layer path: same engine inputs, compare every buffer."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from recnn_amd import _lib as L
from recnn_amd.nn.engine import StepEngine
from tests.test_gpu_engine import _init_nets, _rand_batch
from tests.helpers import rel_err
lib = L.load()
S, A, H = 1290, 128, 256
for algo, B in (("ddpg", 2048), ("td3", 777)):
    nc = 2 if algo == "td3" else 1
    actor, critics = _init_nets(1, S, A, H, nc)
    gen = torch.Generator().manual_seed(2)
    batch = _rand_batch(B, S, A, gen)
    nm = 8 if algo == "td3" else 6
    masks = [(torch.rand(B, H, generator=gen) < 0.5).to(torch.uint8) for _ in range(nm)]
    noise = torch.randn(B, A, generator=gen) * 0.5
    outs = []
    for fused in (0, 1):
        from recnn_amd._tune import set_default_tuning; set_default_tuning(fused_mlp=fused)
        eng = StepEngine(algo, S, A, H, B, dtype="bf16", mask_mode="external")
        nets = [(0, actor), (1, actor), (2, critics[0]), (3, critics[0])] + ([(4, critics[1]), (5, critics[1])] if nc == 2 else [])
        for ni, p in nets: eng.load_params(ni, p)
        eng.set_hyper(policy_opt=dict(lr=1e-3), value_opt=dict(lr=1e-3)); eng.set_counters()
        eng.pack_batch(batch["state"], batch["action"], batch["reward"], batch["next_state"], batch["done"])
        eng.set_external(masks=masks, noise=noise if algo == "td3" else None)
        eng.step(B, True, 0)
        torch.cuda.synchronize()
        names = ["next_action", "gen_action", "expected", "q1", "q_pi", "critic1_h1", "critic1_h2", "actor_h1", "actor_h2", "pc_h1", "pc_h2", "critic1_dz1", "dact"]
        outs.append(({n: eng.buffer(n, B) for n in names}, eng.losses(), {k: v.clone() for k, v in eng.param_views(0).items()}))
    print(algo, B, "losses", outs[0][1], outs[1][1])
    for n in outs[0][0]:
        a, b = outs[0][0][n], outs[1][0][n]
        print("  %-12s max|diff| %.3e  rel %.3e  equal=%s" % (n, float((a - b).abs().max()), rel_err(b, a), torch.equal(a, b)))
    print("  policy w1 equal:", torch.equal(outs[0][2]["w1"], outs[1][2]["w1"]), rel_err(outs[1][2]["w1"], outs[0][2]["w1"]))
