#!/usr/bin/env python
"""Which torch streams share a hardware queue?  Two spin kernels (torch.cuda._sleep) on two streams take T when the streams
sit on different hardware queues and 2T when they share one (HIP multiplexes its streams onto GPU_MAX_HW_QUEUES = 4 HSA
queues; a stream is bound to a queue when it is first used).  Prints the pairwise matrix of finish times in units of T."""
import sys
import torch

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
main = torch.cuda.current_stream()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cands = [torch.cuda.Stream(device=dev) for _ in range(N)]
hp = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(2)]
CYC = 2_000_000
names = ["main"] + [f"s{i}" for i in range(N)] + [f"hp{i}" for i in range(2)]
allst = [main] + cands + hp
for s in allst:                      # first use, in this order
    with torch.cuda.stream(s):
        torch.cuda._sleep(1000)
torch.cuda.synchronize()


def spin_alone(s):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        e0.record(s)
        torch.cuda._sleep(CYC)
        e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for _ in range(3):
    T = spin_alone(main)
print(f"one spin = {T:.2f} ms")


def group(streams):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    ends = []
    e0.record(streams[0])
    for s in streams[1:]:
        s.wait_event(e0)
    for s in streams:
        with torch.cuda.stream(s):
            torch.cuda._sleep(CYC)
            e = torch.cuda.Event(enable_timing=True)
            e.record(s)
            ends.append(e)
    torch.cuda.synchronize()
    return [e0.elapsed_time(e) / T for e in ends]


print("pairs: max finish time in units of one spin (2.0 = the two share a queue)")
print("      " + " ".join(f"{n:>5s}" for n in names))
for i, a in enumerate(allst):
    row = []
    for j, b in enumerate(allst):
        row.append("    ." if j <= i else f"{max(group([a, b])):5.1f}")
    print(f"{names[i]:>5s} " + " ".join(row))
print("all at once: " + " ".join(f"{n}={t:.1f}" for n, t in zip(names, group(allst))))
