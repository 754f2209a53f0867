"""Does the multi-stream schedule make the caching allocator's pool grow?  Reserved / allocated memory every 150 steps."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from scat_amd.trainer import TrainStep  # noqa: E402

dev = torch.device("cuda", 0)
net = bench.make_net("resnet50", 1, dev)
ts = TrainStep(net, lr=5e-4)
u8, lab = bench.build_inputs(96, 100, dev)
from scat_amd import ops as _ops
x = _ops.preprocess_u8(u8, (224, 224))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 900
t0 = time.perf_counter()
for i in range(n + 1):
    ts(x, lab)
    if i % 150 == 0:
        print(f"step {i:5d}: reserved {torch.cuda.memory_reserved() / 2**30:6.2f} GiB, allocated "
              f"{torch.cuda.memory_allocated() / 2**30:5.2f} GiB, peak {torch.cuda.max_memory_allocated() / 2**30:5.2f} GiB",
              flush=True)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / (n + 1) * 1e3:.2f} ms/step")
