"""Every scheduling switch (token stream, side streams, early Adam, prepared weights, folded BatchNorm backward,
packed shortcut input, fused dx2 sum) on versus all of them off: the first train step must give the same loss bit for
bit (same kernels, same summation order inside each), later steps may differ only by fp32 summation-order effects."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
def test_switches_do_not_change_the_numbers():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from tools import ab_check

    a, b = ab_check.run(13, {}), ab_check.run(13, ab_check.OFF)
    # (the first loss used to be bit-identical; with the BatchNorm sums taken in the convolution epilogues — fp32 over
    # each tile's columns, then fp64 — it depends on the tile shape of the kernel that ran, at the 1e-7 level)
    assert abs(a[0] - b[0]) / abs(b[0]) < 2e-6, (a, b)
    for u, v in zip(a[1:], b[1:]):
        assert abs(u - v) / abs(v) < 5e-2, (a, b)


@pytest.mark.timeout(600)
def test_step_streams_do_not_share_hardware_queues():
    """scat_amd.streams picks every role's stream by measuring which pool streams share a hardware queue (HIP multiplexes
    all streams of a process onto four): after a train step the backbone's weight-gradient stream and the token stream run
    beside the main stream and beside each other, and the probe itself tells one stream from two on one queue."""
    import torch

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    from scat_amd import streams

    dev = torch.device("cuda", 0)
    net = bench.make_net("resnet50", 1, dev)
    step = bench.Step("resnet50", net, dev)
    u8, lab = bench.build_inputs(8, 5, dev)
    for _ in range(2):
        step(u8, lab)
    torch.cuda.synchronize()
    roles = streams.bound(dev)
    assert {"wgrad", "tokens"} <= set(roles), list(roles)
    pairs = {frozenset((a, b)) for a, b, _ in streams.sharing(dev)}
    for bad in (("main", "wgrad"), ("main", "tokens"), ("wgrad", "tokens"), ("main", "tokens_wgrad"), ("main", "opt")):
        assert frozenset(bad) not in pairs, pairs
    main = torch.cuda.default_stream(dev)
    assert streams._shares(main, main)
    # the probe finds the queue-mates of a stream among the pool: with four hardware queues, 8 more streams cannot all
    # run beside main, the weight-gradient stream and the token stream
    more = [torch.cuda.Stream(device=dev) for _ in range(8)]
    heavy = [main, roles["wgrad"], roles["tokens"]]
    assert any(any(streams._shares(s, h) for h in heavy) for s in more)


_CAPTURE_CHILD = r"""
import random, sys
import torch
sys.path.insert(0, ".")
import bench
from scat_amd import graphed

config, batch = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda", 0)


def run(graph):
    random.seed(7)
    net = bench.make_net(config, 1, dev)
    net.mask_rate = 0.0
    step = bench.Step(config, net, dev)
    u8, lab = bench.build_inputs(batch, 100, dev)
    gs = graphed.GraphedStep(lambda: step(u8, lab), warmup=3 if graph else 1 << 30)
    for _ in range(4):
        out = gs()
    torch.cuda.synchronize()
    assert (gs.graph is not None) == graph
    return [p.detach().clone() for p in net.parameters()], out[0].detach().clone()


p_eager, l_eager = run(False)
p_graph, l_graph = run(True)
assert torch.isfinite(l_eager).all() and torch.equal(l_eager, l_graph), (l_eager, l_graph)
assert all(torch.equal(a, b) for a, b in zip(p_eager, p_graph))
print("CAPTURE-OK")
"""


@pytest.mark.timeout(600)
@pytest.mark.parametrize("config,batch", [("resnet50", 8), ("hrnet_w32", 4)])
def test_whole_step_is_stream_capturable(config, batch):
    """A whole train step — forward, backward on every side stream, update — can be captured into ONE HIP graph, and the
    replay leaves exactly the bits the eager step leaves (scat_amd/graphed.py: the fork / join graph of the step is a star
    around the calling stream; branch streams never wait on each other, no work on the legacy default stream).  The
    masked-token draw is off: its upload is host work a capture cannot hold.  In a child process: what this guards
    against showed up as segmentation faults inside hipStreamEndCapture, which must not take the test run down."""
    import subprocess

    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    r = subprocess.run([sys.executable, "-c", _CAPTURE_CHILD, config, str(batch)], cwd=root, capture_output=True, text=True,
                       timeout=550)
    assert r.returncode == 0 and "CAPTURE-OK" in r.stdout, (r.returncode, r.stdout[-400:], r.stderr[-1200:])
