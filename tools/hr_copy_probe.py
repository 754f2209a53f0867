#!/usr/bin/env python
"""Where do the device-to-device copies and the torch `add` kernels of an HRNet-W32 train step come from?  Two steps under
torch.profiler with Python stacks: every aten::copy_ / clone / contiguous / add / add_ / cat / zeros call with the scat_amd
source line that issued it."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
net = bench.make_net("hrnet_w32", 1, dev)
step = bench.Step("hrnet_w32", net, dev)
u8, lab = bench.build_inputs(96, 100, dev)
for _ in range(3):
    step(u8, lab)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(2):
        step(u8, lab)
    torch.cuda.synchronize()
want = {"aten::copy_", "aten::clone", "aten::contiguous", "aten::add", "aten::add_", "aten::cat", "aten::zeros",
        "aten::zeros_like", "aten::fill_", "aten::zero_", "aten::mul", "aten::_foreach_copy_", "aten::_foreach_add_"}
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in want:
        frames = [f for f in (ev.stack or []) if "scat_amd" in f or "bench.py" in f]
        where = frames[0].strip() if frames else ("<autograd engine / torch internals>" if not ev.stack else ev.stack[0].strip())
        numel = 1
        for s in (ev.input_shapes[0] if ev.input_shapes and ev.input_shapes[0] else []):
            numel *= s
        cnt[(ev.name, where[:110], numel >= 1 << 16)] += 1
allc = collections.Counter(ev.name for ev in prof.events())
print("all ops per 2 steps:", allc.most_common(45))
evs = list(prof.events())
mem = [e for e in evs if "emcpy" in e.name and e.device_type == torch.autograd.DeviceType.CPU]
print("memcpy runtime calls per 2 steps:", len(mem))
par = collections.Counter()
for m in mem:
    chain = []
    p_ = m.cpu_parent
    while p_ is not None and len(chain) < 4:
        chain.append(p_.name[:60])
        p_ = p_.cpu_parent
    par[(m.name, " < ".join(chain))] += 1
for (n, ch), c in par.most_common(12):
    print(f"{c:6d}  {n}  inside  {ch}")
print("calls per 2 steps | op | big tensor | issued from")
for (name, where, big), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{n:6d}  {name:22s} {'big' if big else 'small':5s}  {where}")
