"""Can one data-parallel cycle (phase graphs + RCCL all-reduces) be captured into ONE graph?  Probes, at world size 1 (nccl):
(1) dist.all_reduce under torch.cuda.graph capture, (2) hipGraphLaunch of the engine's phase graphs into a capturing stream,
(3) replay timing against the uncaptured loop.   usage: python tools/dp_capture_probe.py"""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import recnn_amd
from recnn_amd.nn import fused
from recnn_amd.parallel import DataParallelStepper

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
items, ratings, off, lens = bench.synthetic_store(0)
table = torch.randn(bench.N_ITEMS, bench.EMB, generator=torch.Generator().manual_seed(0))
env = recnn_amd.data.env.FrameEnv.from_store(table, items, ratings, off, frame_size=10, batch_size=25, device=dev, test_fraction=0.0)
fused.set_defaults(dtype="bf16", mask_mode="hash", seed=1)
recnn_amd.nn.algo.set_default_optimizer("adam")
torch.manual_seed(0)
algo = recnn_amd.nn.DDPG(recnn_amd.nn.Actor(1290, 128, 256, 6e-1), recnn_amd.nn.Critic(1290, 128, 256, 54e-2)).to(dev)
algo.attach_env(env, rows_per_batch=2048, users_per_batch=256)
eng = algo._fused_ctx.engine
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    dp = DataParallelStepper(eng, 2048, always_reduce=True)
    dp.run(0, 20)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); dp.run(20, 200); torch.cuda.synchronize()
    print(f"uncaptured loop: {(time.perf_counter() - t0) * 1e6 / 200:.1f} us/step")
    # (1) all-reduce alone under capture
    x = torch.ones(1 << 18, device=dev)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=stream):
            dist.all_reduce(x)
        g.replay(); torch.cuda.synchronize()
        print("capture of dist.all_reduce: ok, x[0] =", float(x[0]))
    except Exception as ex:
        print("capture of dist.all_reduce FAILED:", type(ex).__name__, str(ex)[:300])
        sys.exit(0)
    # (2) + (3) one policy cycle of the stepper
    g2 = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g2, stream=stream):
            dp.run(220, 10)
        print("capture of a 10-step cycle: ok")
    except Exception as ex:
        print("capture of a cycle FAILED:", type(ex).__name__, str(ex)[:300])
        sys.exit(0)
    for _ in range(3):
        g2.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g2.replay()
    torch.cuda.synchronize()
    print(f"captured cycles: {(time.perf_counter() - t0) * 1e6 / 200:.1f} us/step; losses {eng.losses()}")
