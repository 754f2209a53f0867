"""cProfile of the host side of one bench.Step (where do the ~20 us per launch go?)."""
import cProfile, os, pstats, sys, io
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "hrnet_w32"
dev = torch.device("cuda", 0)
net = bench.make_net(cfg, 1, dev)
step = bench.Step(cfg, net, dev)
u8, lab = bench.build_inputs(96, 100, dev)
for _ in range(3):
    step(u8, lab)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step(u8, lab)
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
