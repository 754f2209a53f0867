"""Data parallelism on the real model, two ranks on ONE MI355X (gloo carries the collectives; RCCL refuses two
ranks on one device).  The collectives take the ASYNCHRONOUS route of the RCCL path: each bucket is snapshotted on the
stream its all-reduce is ordered after (scat_amd.dp.GradBuckets._ordered_stream: the caller's stream + every registered
producer stream, by events) while the backward is still running, and reduced when the step waits for its collectives;
with SCAT_DP_CHECK=1 the flat gradient buffer is poisoned with NaN before every backward and every bucket is tested on
that stream in front of its collective.  Gradients are held to golden G9 (the reference's own two-shard average).  Unattended mode — nothing but the reference's train.py loop (forward, backward, plain
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


def _check_against_g9(named_grads, head_tol=5e-4):
    """Hold averaged gradients to golden G9 (tests/golden/dp.npz: two shards through the REAL reference network,
    gradients averaged — SURVEY §8e's parity definition).  Head parameters (well conditioned) norm-wise at 5e-4 on the
    values the golden stores in full and on the digests; backbone parameters at the noise level of this random-weight
    network's own fp32 gradients (CPU fp32 vs fp64 differ by 2-9 % here), i.e. the L2 norm of every
    averaged gradient within 25 % — a missing 1/N, an unreduced bucket or a wrong shard shows up as >= 50 %."""
    import numpy as np

    from oracle.util import digest, digest_err, rel_err

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dp.npz"))
    worst_head, worst_bb = 0.0, 0.0
    for n, grad in named_grads:
        gc = grad.detach().float().cpu()
        if n.startswith("main_encoder."):
            ref_norm = float(g["gnorm:" + n][0])
            e = abs(float(gc.norm()) - ref_norm) / max(ref_norm, 1e-30)
            worst_bb = max(worst_bb, e)
            assert e < 0.25, (n, e)
        else:
            e = digest_err(digest(gc, 8), g["g:" + n])
            if "g_full:" + n in g.files:
                e = max(e, rel_err(gc, g["g_full:" + n]))
            worst_head = max(worst_head, e)
            assert e < head_tol, (n, e)
    return worst_head, worst_bb


def _worker(rank, world, port, out):
    try:
        import numpy as np
        import torch.distributed as dist

        from scat_amd import synth
        from tests.test_gpu_model import make_encoder

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                          LOCAL_RANK="0", SCAT_DIST_BACKEND="gloo")
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        from scat_amd.trainer import scat_loss

        torch.manual_seed(1234 + rank)               # (ranks would start from different weights...
        net = make_encoder(43)                       # ...this test loads identical ones; the attach broadcasts anyway)
        opt = torch.optim.Adam(net.parameters(), lr=1e-4)
        x = T(synth.images(700 + rank, 4)).cuda()    # each rank its own shard: the shards of golden G9
        lab = T(synth.labels(710 + rank, 4)).cuda()
        # local (non-DP) gradients of this shard from a twin with the same weights, with WORLD_SIZE hidden
        os.environ["WORLD_SIZE"] = "1"
        twin = make_encoder(43)
        random.seed(11)
        scat_loss(twin(x)[0], lab)[0].backward()
        os.environ["WORLD_SIZE"] = str(world)
        random.seed(11)
        scat_loss(net(x)[0], lab)[0].backward()      # first forward attaches the buckets (dist via gloo)
        assert net._dp_buckets.world == world
        wh, wb = _check_against_g9([(n, p.grad) for n, p in net.named_parameters()])
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
                          LOCAL_RANK="0", SCAT_DIST_BACKEND="gloo", SCAT_DP_CHECK="1")
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
        assert ts.buckets.check                               # SCAT_DP_CHECK=1: no bucket left before it was written
        _check_against_g9([(n, ts.buckets.view_for(p)) for n, p in net.named_parameters()])
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


def _worker_rccl_single(port, out):
    """ONE rank, backend nccl (= RCCL), SCAT_DP_FORCE_COLLECTIVES=1: everything the N > 1 path does on hardware except
    talk to a peer — init_process_group(device_id=...), the replica broadcast, the synchronous ReduceOp.AVG probe,
    all_reduce(async_op=True) per bucket with the private ordering stream current, work.wait() on the optimiser stream
    (early Adam inside the backward), finish() — with the self-check on.  A one-rank AVG is the identity, so the
    gradients and the Adam-updated weights must equal a plain single-process twin BIT FOR BIT."""
    try:
        import numpy as np
        import torch.distributed as dist

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
                          SCAT_DP_CHECK="1", SCAT_DP_FORCE_COLLECTIVES="1")
        os.environ.pop("SCAT_DIST_BACKEND", None)
        from scat_amd import synth
        from scat_amd.dp import init_distributed
        from scat_amd.trainer import TrainStep
        from tests.test_gpu_model import make_encoder

        assert not torch.cuda.is_initialized()       # init_distributed() must come before any GPU call of the process
        rank, local, world = init_distributed()
        assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        x, lab = T(synth.images(700, 4)).cuda(), T(synth.labels(710, 4)).cuda()
        net = make_encoder(43)
        net.train()
        ts = TrainStep(net, lr=1e-4)
        b = ts.buckets
        assert b.collective and b.world == 1 and b.check
        probed = b._avg_ok                           # RCCL of ROCm 7 has ReduceOp.AVG; the SUM fallback scales by 1/1
        issued = []
        orig = dist.all_reduce

        def spy(t, *a, **k):
            issued.append((t.numel(), bool(k.get("async_op", False)), torch.cuda.current_stream() == b._comm_stream))
            return orig(t, *a, **k)

        dist.all_reduce = spy
        try:
            random.seed(11)
            ts(x, lab)
            random.seed(12)
            ts(x, lab)                               # a second step: buffers recycled, early Adam waits again
        finally:
            dist.all_reduce = orig
        torch.cuda.synchronize()
        # 7 buckets per step, all asynchronous, all issued with the ordering stream current
        assert len(issued) == 2 * len(b.ranges), issued
        assert all(a and on_c for _, a, on_c in issued), issued
        assert sorted(n for n, _, _ in issued[:7]) == sorted(e - a for a, e in b.ranges.values())
        got_g, got_p = b.flat_grad.detach().cpu().clone(), b.flat_param.detach().cpu().clone()
        # the twin: same process, no collectives at all
        os.environ["SCAT_DP_FORCE_COLLECTIVES"] = "0"
        twin = make_encoder(43)
        twin.train()
        tts = TrainStep(twin, lr=1e-4)
        assert not tts.buckets.collective
        random.seed(11)
        tts(x, lab)
        random.seed(12)
        tts(x, lab)
        torch.cuda.synchronize()
        assert torch.equal(got_g, tts.buckets.flat_grad.detach().cpu())
        assert torch.equal(got_p, tts.buckets.flat_param.detach().cpu())
        out.put((0, "OK avg=%s" % probed))
        dist.destroy_process_group()
    except Exception as e:   # noqa: BLE001
        import traceback
        out.put((0, "ERR " + repr(e) + "\n" + traceback.format_exc()))


@pytest.mark.timeout(600)
def test_rccl_single_rank_collectives():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_rccl_single, args=(_free_port(), q))
    p.start()
    got = q.get(timeout=500)
    p.join(60)
    assert isinstance(got[1], str) and got[1].startswith("OK"), got


@pytest.mark.timeout(900)
def test_bench_two_ranks_through_the_launcher():
    """``bench.py --gpus 2`` exactly as the driver launches it for N > 1 (python -m torch.distributed.run --nnodes=1
    --nproc-per-node 2 --master-addr 127.0.0.1 ...), both ranks on the one GPU of this box with gloo carrying the
    collectives (RCCL refuses two ranks on one device): the launcher env is read, the process group comes up before any GPU
    call, every rank takes its own shard, the barriers / MAX-reduce of the wall time / gathered per-rank evidence run,
    rank 0 prints ONE JSON line for the whole job.  (this branch of bench.py had never executed.)"""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCAT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 192 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and abs(d["value"] - 192 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-2 * d["value"]
    import math
    assert math.isfinite(d["config"]["final_loss"])
    dp = d["config"]["data_parallel"]
    assert dp["world"] == 2 and dp["backend"] == "gloo" and dp["collectives_per_step"] == 7
    assert sorted(x["rank"] for x in dp["per_rank"]) == [0, 1]
    assert sum(dp["bucket_bytes"].values()) >= 4 * 29_401_307
    assert d["roofline"] is not None and d["roofline"]["frac"] > 0
