"""Data-parallel gradient buckets over torch.distributed, world_size 2, gloo on CPU.
Checks the N>1 logic of scat_amd.dp.GradBuckets (bucket order, flat views, averaging, finish()) and the
DP parity definition of SURVEY §8(e): N replicas == mean of the per-shard gradients, one Adam step."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp
import torch.nn as nn


class _Tiny(nn.Module):
    """parameter names shaped like the real model's, so bucket assignment is exercised"""

    def __init__(self):
        super().__init__()
        self.main_encoder = nn.ModuleDict({
            "conv1": nn.Linear(6, 5, bias=False), "layer1": nn.Linear(5, 5), "layer2": nn.Linear(5, 5),
            "layer3": nn.Linear(5, 5), "layer4": nn.Linear(5, 5), "fc1": nn.Linear(5, 4)})
        self.regressor = nn.Linear(4, 3)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, 3))

    def forward(self, x):
        e = self.main_encoder
        h = torch.tanh(e["conv1"](x))
        for k in ("layer1", "layer2", "layer3", "layer4"):
            h = torch.tanh(e[k](h))
        return self.regressor(torch.tanh(e["fc1"](h))) + self.mask_token.view(1, 3)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import torch.distributed as dist

    from scat_amd.dp import BACKBONE_BUCKETS, GradBuckets

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = _Tiny()                      # identical weights on both ranks
    ref = _Tiny()
    ref.load_state_dict(net.state_dict())
    b = GradBuckets(net)
    assert list(b.ranges) == ["head", *BACKBONE_BUCKETS]
    # parameters alias the flat buffer
    p0 = net.regressor.weight
    off, k = b.slot[p0]
    assert p0.data_ptr() == b.flat_param[off:off + k].data_ptr()
    torch.manual_seed(100 + rank)      # each rank its own shard
    x, y = torch.randn(8, 6), torch.randn(8, 3)
    loss = (net(x) - y).square().mean()
    loss.backward()
    # replay the fused-backward protocol: head first, then stage buckets in backward order
    bb = {n: p for n, p in net.named_parameters() if n.startswith("main_encoder.")}
    saved = {n: p.grad.clone() for n, p in bb.items()}
    for p in bb.values():
        p.grad = None
    b.begin_backbone()
    for bucket in BACKBONE_BUCKETS:
        for n, p in bb.items():
            key = n.split(".")[1]
            if (key == bucket) or (bucket == "stem" and key == "conv1"):
                b.view_for(p).copy_(saved[n])
        b.ready((bucket,))
    b.adopt(list(bb.values()))
    b.finish()
    # reference: gather every rank's plain gradients and average
    ref_loss = (ref(x) - y).square().mean()
    ref_loss.backward()
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        g = q.grad.clone()
        dist.all_reduce(g)
        g /= world
        assert torch.allclose(p.grad, g, atol=1e-7), n
        assert p.grad.data_ptr() == b.view_for(p).data_ptr(), n   # grads live in the flat bucket
    # one Adam step on the averaged gradients keeps replicas bit-identical
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    opt.step()
    flat = b.flat_param.clone()
    other = flat.clone()
    dist.broadcast(other, src=0)
    assert torch.equal(flat, other)
    b.zero_grad()
    assert all(p.grad is None for p in net.parameters())
    out.put((rank, float(loss)))
    dist.destroy_process_group()


def _worker_auto(rank, world, port, out):
    """unattended mode: nothing but forward / backward / optimizer.step(), like the reference's train.py"""
    import torch.distributed as dist

    from scat_amd import dp

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank), SCAT_DIST_BACKEND="gloo")
    torch.manual_seed(0)
    net, ref = _Tiny(), _Tiny()
    ref.load_state_dict(net.state_dict())
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)          # built BEFORE the buckets exist, as train.py does
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    for step in range(3):
        dp.auto_attach(net)                                    # what EncoderTransformer.forward does
        torch.manual_seed(100 * step + rank)
        x, y = torch.randn(8, 6), torch.randn(8, 3)
        opt.zero_grad()
        (net(x) - y).square().mean().backward()
        ropt.zero_grad()
        (ref(x) - y).square().mean().backward()
        for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
            g = q.grad.clone()
            dist.all_reduce(g)
            g /= world
            q.grad = g                                         # the reference, averaged by hand
            assert torch.allclose(p.grad, g, atol=1e-7), (step, n)
        opt.step()
        ropt.step()
        for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
            assert torch.allclose(p, q, atol=1e-7), (step, n)
    assert net._dp_buckets.world == world
    out.put((rank, 0.0))
    dist.destroy_process_group()


def _worker_diverged_start(rank, world, port, out):
    """ranks construct the model from DIFFERENT seeds (the reference's train.py never seeds): attaching the buckets
    must leave every rank with rank 0's parameters and buffers, as DistributedDataParallel does at construction;
    a backbone without the fused-backward protocol must still get all of its buckets gathered and reduced"""
    import torch.distributed as dist

    from scat_amd.dp import GradBuckets

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1000 + rank)
    net = _Tiny()
    net.register_buffer("running_mean", torch.randn(7))
    net.register_buffer("num_batches_tracked", torch.tensor(3 + rank, dtype=torch.long))
    mine = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    b = GradBuckets(net)
    flat = b.flat_param.clone()
    other = flat.clone()
    dist.broadcast(other, src=0)
    assert torch.equal(flat, other)
    now = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.equal(now, mine) == (rank == 0)       # rank 1's own draw is gone
    for buf in (net.running_mean, net.num_batches_tracked.float()):
        ref = buf.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(buf, ref)
    assert int(net.num_batches_tracked) == 3
    # plain autograd backbone (no ready()/adopt()): finish() gathers and reduces every bucket
    torch.manual_seed(50 + rank)
    x, y = torch.randn(8, 6), torch.randn(8, 3)
    (net(x) - y).square().mean().backward()
    local = {n: p.grad.clone() for n, p in net.named_parameters()}
    b.finish()
    for n, p in net.named_parameters():
        g = local[n]
        dist.all_reduce(g)
        g /= world
        assert torch.allclose(p.grad, g, atol=1e-7), n
        assert p.grad.data_ptr() == b.view_for(p).data_ptr(), n
    frozen = torch.nn.Parameter(torch.zeros(2), requires_grad=False)
    assert b.view_for(frozen) is None
    out.put((rank, 0.0))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_replicas_start_from_rank0():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_diverged_start, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
        assert p.exitcode == 0
    assert sorted(r for r, _ in (q.get(timeout=5) for _ in range(2))) == [0, 1]


@pytest.mark.timeout(120)
def test_two_rank_unattended():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_auto, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
        assert p.exitcode == 0
    assert sorted(r for r, _ in (q.get(timeout=5) for _ in range(2))) == [0, 1]


@pytest.mark.timeout(120)
def test_two_rank_buckets():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert [r for r, _ in got] == [0, 1]


def test_stream_order_self_check(monkeypatch):
    """SCAT_DP_CHECK=1: the flat gradient buffer is poisoned with NaN before every backward and every bucket is tested
    in front of its collective — a bucket declared ready before all of its slices were written is reported (this is
    how a gradient kernel on a stream nobody joined shows up); the alignment padding is nobody's to write."""
    from scat_amd.dp import BACKBONE_BUCKETS, GradBuckets

    monkeypatch.setenv("SCAT_DP_CHECK", "1")
    net = _Tiny()
    b = GradBuckets(net)
    assert b.check
    for trial in range(2):
        b.zero_grad()
        assert torch.isnan(b.view_for(net.regressor.weight)).all()
        (net(torch.randn(4, 6)).sum()).backward()
        bb = {n: p for n, p in net.named_parameters() if n.startswith("main_encoder.")}
        saved = {n: p.grad.clone() for n, p in bb.items()}
        for p in bb.values():
            p.grad = None
        b.begin_backbone()
        for bucket in BACKBONE_BUCKETS:
            for n, p in bb.items():
                key = n.split(".")[1]
                if (key == bucket) or (bucket == "stem" and key == "conv1"):
                    if trial == 1 and n == "main_encoder.layer3.bias":
                        continue                  # "its kernel has not run yet"
                    b.view_for(p).copy_(saved[n])
            b.ready((bucket,))
        b.adopt(list(bb.values()))
        if trial == 0:
            b.finish()
        else:
            with pytest.raises(RuntimeError, match="layer3"):
                b.finish()


def test_single_process_buckets_are_storage_only():
    from scat_amd.dp import GradBuckets

    net = _Tiny()
    b = GradBuckets(net)
    assert b.world == 1
    (net(torch.randn(4, 6)).sum()).backward()
    b.finish()     # gathers every bucket nobody declared ready, no collective
    for p in net.parameters():
        assert p.grad.data_ptr() == b.view_for(p).data_ptr()
    assert net.regressor.weight.grad.data_ptr() == b.view_for(net.regressor.weight).data_ptr()
    off, k = b.slot[net.regressor.weight]
    assert off % 64 == 0     # 256-B aligned slots for the kernels' 16-B loads
