#!/usr/bin/env python
"""Does it pay to confine the backbone's weight-gradient stream to a subset of the CUs (hipExtStreamCreateWithCUMask), so that
the main stream's dependent chain of kernels keeps the rest to itself?  One process per variant (the stream plan is made
once): argv[1] = number of CUs the side stream may use (0: the planned pool stream, the shipped configuration)."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from scat_amd import streams  # noqa: E402

ncu = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda", 0)
torch.cuda.init()
net = bench.make_net("resnet50", 1, dev)
step = bench.Step("resnet50", net, dev)
u8, lab = bench.build_inputs(96, 100, dev)
if ncu > 0:
    hip = ctypes.CDLL("libamdhip64.so")
    total = torch.cuda.get_device_properties(0).multi_processor_count
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    # spread the allowed CUs evenly over the index space (the runtime interleaves shader engines / XCDs over the bits)
    k = 0
    for i in range(total):
        if (i * ncu) // total != ((i + 1) * ncu) // total:
            mask[i // 32] |= 1 << (i % 32)
            k += 1
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), words, mask)
    assert rc == 0 and st.value, rc
    ext = torch.cuda.ExternalStream(st.value, device=dev)
    streams.get(dev, "wgrad")               # make the plan, then replace the role's stream
    streams.bound(dev)["wgrad"] = ext
    print(f"weight-gradient stream confined to {k} of {total} CUs", flush=True)
for _ in range(6):
    step(u8, lab)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        step(u8, lab)
    torch.cuda.synchronize()
    print(f"side CUs {ncu or 'all (planned stream)'}: {1e3 * (time.perf_counter() - t0) / 20:.2f} ms/step", flush=True)
