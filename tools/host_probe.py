"""How far ahead of the GPU does the Python side run?  Times the enqueue of a train step (no synchronisation)
against the synchronised step: if enqueue << step the path is GPU-bound and launch capture (hipGraph) has nothing
to win; if they are close the host is the limit."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=96)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--config", default="resnet50", choices=["resnet50", "hrnet_w32", "performer"])
    ap.add_argument("--sync-debug", action="store_true")
    ap.add_argument("--hi", action="store_true", help="run the step on a high-priority stream (side streams stay normal)")
    a = ap.parse_args()
    from scat_amd.trainer import TrainStep

    dev = torch.device("cuda", 0)
    net = bench.make_net(a.config, 1, dev)
    ts = TrainStep(net, lr=5e-4)
    u8, lab = bench.build_inputs(a.batch, 100, dev)
    from scat_amd import ops as _ops
    x = _ops.preprocess_u8(u8, (224, 224))
    if a.hi:
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
    for _ in range(5):
        ts(x, lab)
    torch.cuda.synchronize()
    if a.sync_debug:            # torch warns at every call that makes the host wait for the device
        torch.cuda.set_sync_debug_mode("warn")
        ts(x, lab)
        torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()
    enq = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        t = time.perf_counter()
        ts(x, lab)
        enq.append(time.perf_counter() - t)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    enq.sort()
    print(f"batch {a.batch}: enqueue median {1e3 * enq[len(enq) // 2]:.2f} ms/step (min {1e3 * enq[0]:.2f}), "
          f"all enqueued after {1e3 * t_enq:.1f} ms, finished after {1e3 * t_all:.1f} ms "
          f"({1e3 * t_all / a.steps:.2f} ms/step)")


if __name__ == "__main__":
    main()
