"""Host enqueue time vs GPU time of one bench.Step for a config (is the step launch-bound?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "hrnet_w32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
net = bench.make_net(cfg, 1, dev)
step = bench.Step(cfg, net, dev)
u8, lab = bench.build_inputs(96, 100, dev)
for _ in range(3):
    step(u8, lab)
torch.cuda.synchronize()
enq = []
t0 = time.perf_counter()
for _ in range(steps):
    t = time.perf_counter()
    step(u8, lab)
    enq.append(time.perf_counter() - t)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
enq.sort()
print(f"{cfg}: enqueue median {1e3 * enq[len(enq) // 2]:.2f} ms/step (min {1e3 * enq[0]:.2f}), all enqueued after "
      f"{1e3 * t_enq:.1f} ms, finished after {1e3 * t_all:.1f} ms ({1e3 * t_all / steps:.2f} ms/step)")
