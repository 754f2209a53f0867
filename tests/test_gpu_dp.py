"""Data parallelism on the real model, two ranks on ONE MI355X (gloo carries the collectives; RCCL refuses two
ranks on one device).  Unattended mode — nothing but the reference's train.py loop (forward, backward, plain
torch.optim.Adam) under WORLD_SIZE=2 — must give every rank the MEAN of the per-rank gradients."""
import os
import random
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    try:
        import numpy as np
        import torch.distributed as dist

        from scat_amd import synth
        from tests.test_gpu_model import make_encoder

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                          LOCAL_RANK="0", SCAT_DIST_BACKEND="gloo")
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        net = make_encoder(41)                       # identical weights on both ranks
        opt = torch.optim.Adam(net.parameters(), lr=1e-4)
        x = T(synth.images(500 + rank, 2)).cuda()    # each rank its own shard
        cot = T(synth.normal_like(600 + rank, "cot", (2, 66))).cuda()
        # local (non-DP) gradients of this shard from a twin with the same weights, with WORLD_SIZE hidden
        os.environ["WORLD_SIZE"] = "1"
        twin = make_encoder(41)
        random.seed(7)
        (twin(x)[0] * cot).sum().backward()
        os.environ["WORLD_SIZE"] = str(world)
        random.seed(7)
        (net(x)[0] * cot).sum().backward()           # first forward attaches the buckets (dist via gloo)
        assert net._dp_buckets.world == world
        worst, bad = 0.0, []
        for (n, p), (_, q) in zip(net.named_parameters(), twin.named_parameters()):
            g = q.grad.detach().cpu()
            dist.all_reduce(g)
            g /= world
            err = (p.grad.cpu() - g).abs().max().item() / (g.abs().max().item() + 1e-20)
            worst = max(worst, err)
            if err >= 1e-5:
                bad.append((n, round(err, 6)))
            assert p.grad.data_ptr() == net._dp_buckets.view_for(p).data_ptr(), n
        assert not bad, (len(bad), bad[:12])
        opt.step()
        flat = net._dp_buckets.flat_param.detach().cpu()
        other = flat.clone()
        dist.broadcast(other, src=0)
        assert torch.equal(flat, other)              # replicas stay bit-identical after the step
        out.put((rank, worst))
        dist.destroy_process_group()
    except Exception as e:   # noqa: BLE001 - report to the parent instead of hanging its join
        import traceback
        out.put((rank, "ERR " + repr(e) + "\n" + traceback.format_exc()))


@pytest.mark.timeout(600)
def test_unattended_two_ranks_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=500) for _ in range(2)]
    for p in procs:
        p.join(60)
    assert all(not isinstance(w, str) for _, w in got), got
    assert sorted(r for r, _ in got) == [0, 1]


def _worker_trainstep(rank, world, port, out):
    """bench.py's path: scat_amd.trainer.TrainStep (explicit buckets, fused Adam) with the backbone split in two nodes
    and the token path on its own stream — the head bucket is then reduced in the middle of the backward."""
    try:
        import numpy as np
        import torch.distributed as dist

        from scat_amd import synth
        from scat_amd.dp import init_distributed
        from scat_amd.trainer import TrainStep
        from tests.test_gpu_model import make_encoder

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                          LOCAL_RANK="0", SCAT_DIST_BACKEND="gloo")
        init_distributed()
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        x = T(synth.images(700 + rank, 4)).cuda()            # each rank its own shard
        lab = T(synth.labels(710 + rank, 4)).cuda()
        # local gradients of this shard: a twin with the same weights in a world of its own
        twin = make_encoder(43)
        twin.train()
        solo = None
        for r in range(world):                                # (new_group is collective: same order on every rank)
            grp = dist.new_group([r])
            if r == rank:
                solo = grp
        tts = TrainStep(twin, lr=1e-4, process_group=solo)
        random.seed(11)
        tts(x, lab)
        local = tts.buckets.flat_grad.detach().cpu().clone()
        dist.all_reduce(local)
        local /= world
        net = make_encoder(43)
        net.train()
        ts = TrainStep(net, lr=1e-4)
        assert ts.buckets.world == world
        random.seed(11)
        ts(x, lab)
        got = ts.buckets.flat_grad.detach().cpu()
        worst = 0.0
        for name, (a, b) in ts.buckets.ranges.items():
            ref = local[a:b]
            err = (got[a:b] - ref).abs().max().item() / (ref.abs().max().item() + 1e-20)
            worst = max(worst, err)
            assert err < 1e-5, (name, err)
        flat = ts.buckets.flat_param.detach().cpu()
        other = flat.clone()
        dist.broadcast(other, src=0)
        assert torch.equal(flat, other)                       # replicas stay bit-identical after the fused Adam step
        out.put((rank, worst))
        dist.destroy_process_group()
    except Exception as e:   # noqa: BLE001
        import traceback
        out.put((rank, "ERR " + repr(e) + "\n" + traceback.format_exc()))


@pytest.mark.timeout(600)
def test_trainstep_two_ranks_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_trainstep, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=500) for _ in range(2)]
    for p in procs:
        p.join(60)
    assert all(not isinstance(w, str) for _, w in got), got
    assert sorted(r for r, _ in got) == [0, 1]
