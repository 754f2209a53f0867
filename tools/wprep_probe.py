"""Host cost of ops.WeightPrep.run() on the ResNet-50 backbone (106 entries): time per call, enqueue only."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from scat_amd import ops  # noqa: E402
from scat_amd.trainer import TrainStep  # noqa: E402

dev = torch.device("cuda", 0)
net = bench.make_net("resnet50", 1, dev)
ts = TrainStep(net, lr=5e-4)
u8, lab = bench.build_inputs(8, 100, dev)
from scat_amd import ops as _ops
x = _ops.preprocess_u8(u8, (224, 224))
for _ in range(3):
    ts(x, lab)
torch.cuda.synchronize()
wp = net.main_encoder._wprep
print("entries", len(wp.entries), "jobs", wp.table[1], "blocks", wp.table[2])
for label, fn in (("run(True)", lambda: wp.run(True)),
                  ("dead+version scan", lambda: (sum(e[3]._version for e in wp.entries.values()),
                                                 [k for k, e in wp.entries.items() if e[3].data_ptr() != k[0]])),
                  ("launch only", lambda: ops.lib().scat_wprep_run(wp.table[0].data_ptr(), wp.table[1], wp.table[2],
                                                                   torch.cuda.current_stream().cuda_stream))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{label:20s} host {1e6 * (t1 - t0) / 50:8.1f} us/call, incl. GPU {1e6 * (t2 - t0) / 50:8.1f} us/call")
