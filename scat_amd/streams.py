"""Role -> HIP stream, placed on the hardware queues by measurement.

HIP multiplexes the streams of a process onto GPU_MAX_HW_QUEUES = 4 HSA queues: the null stream has one, every stream
created afterwards is bound to one of the others or, from the fourth on, shares one (``tools/queue_probe.py``; torch
creates its pool of 32 streams per device at once, so WHICH pool stream sits on which queue depends on what else drew
from the pool before).  Two streams on one queue are NOT independent: the queue is in-order, so a kernel — or a wait for
an all-reduce, or for another stream's event — recorded on one of them holds back everything the other enqueued later.
Unplanned, the data-parallel path paid 3.8 ms of a 23.6 ms step for that on a one-rank RCCL group: RCCL's streams were
bound first and the backbone's weight-gradient stream ended up on the MAIN queue (``profiles/r03_dp_queues.txt``).  More
queues are not the answer (GPU_MAX_HW_QUEUES >= 5: 31 ms steps).

So the streams of a train step are requested here by role, and a role's stream is CHOSEN: pool streams are drawn until
one is found that does not share a queue with the roles it must run beside (two ~0.5 ms spin kernels finish together or
one after the other: ``_shares``).  What may share is what is sequential anyway:

    main (null stream)
    wgrad          backbone weight gradients                      never with main
    tokens         token path                                     never with main, wgrad
    comm           the stream collectives are ordered after       never with main, wgrad, tokens; with RCCL's stream
    tokens_wgrad   the token path's weight gradients              never with main, wgrad
    aux            pose-length side computation (at the loss)     never with main
    opt            early optimiser update                         never with main, wgrad (= comm when there are collectives)
"""
from __future__ import annotations

import os

import torch

from . import _switches as _sw

PROBE = _sw.ab("SCAT_STREAMS_PROBE", True)   # 0: take pool streams as they come (A/B runs)

_BOUND = {}          # device index -> {role: stream}
ORDER = ("wgrad", "tokens", "comm", "tokens_wgrad", "aux", "opt")
_AVOID = {"wgrad": ("main",), "tokens": ("main", "wgrad"), "comm": ("main", "wgrad", "tokens"),
          "tokens_wgrad": ("main", "wgrad"), "aux": ("main",), "opt": ("main", "wgrad")}
_SPIN = 1_000_000    # cycles (~0.5 ms)


def _dev(device):
    d = torch.device(device)
    return torch.device("cuda", torch.cuda.current_device() if d.index is None else d.index)


def bound(device):
    return _BOUND.setdefault(_dev(device).index, {})


def _finish_times(x, y):
    dev = x.device
    torch.cuda.synchronize(dev)
    e0 = torch.cuda.Event(enable_timing=True)
    ends = []
    e0.record(x)
    y.wait_event(e0)
    for s in (x, y):
        with torch.cuda.stream(s):
            torch.cuda._sleep(_SPIN)
            e = torch.cuda.Event(enable_timing=True)
            e.record(s)
            ends.append(e)
    torch.cuda.synchronize(dev)
    return max(e0.elapsed_time(e) for e in ends)


_ONE = {}
_RETRIES = 4


def _shares(a, b):
    """do streams a and b sit on the same hardware queue?  (device-synchronising; start-up and tests only)"""
    if a is b or a.cuda_stream == b.cuda_stream:
        return True
    key = a.device.index
    if key not in _ONE:      # two spins on ONE stream, after a warm-up
        _finish_times(a, a)
        _ONE[key] = min(_finish_times(a, a) for _ in range(3)) / 2
    # two streams on one queue can never finish both spins in one spin's time, so a single fast measurement proves they
    # run side by side; a slow one may be a clock ramp or a busy host: ask again (the minimum decides)
    # (every "shares" verdict is therefore a minimum over _RETRIES + 1 measurements, every "independent" one a proof)
    t = _finish_times(a, b)
    for _ in range(_RETRIES):
        if t <= 1.35 * _ONE[key]:
            break
        t = min(t, _finish_times(a, b))
    return t > 1.55 * _ONE[key]


def _pick(device, avoid):
    """a pool stream that shares a queue with none of ``avoid`` (the last one drawn if 32 draws find none)"""
    s = None
    for _ in range(32):
        s = torch.cuda.Stream(device=device)
        if not PROBE or not any(_shares(s, a) for a in avoid):
            return s
    return s


def _main(device):
    return torch.cuda.default_stream(_dev(device))


_WARNED = [False]


def bind(device, roles, priority=0):
    """Create the streams of ``roles`` (those not bound yet), in the order given."""
    device = _dev(device)
    b = bound(device)
    if not _WARNED[0]:
        _WARNED[0] = True
        q = os.environ.get("GPU_MAX_HW_QUEUES", "")
        if q.isdigit() and int(q) > 4:
            import warnings
            warnings.warn(f"scat_amd: GPU_MAX_HW_QUEUES={q}: with more than four hardware queues per process the train step "
                          "was measured 30 % slower on MI355X (profiles/r03_dp_queues.txt); unset it", RuntimeWarning)
    for r in roles:
        if r in b:
            continue
        if priority != 0:
            b[r] = torch.cuda.Stream(device=device, priority=priority)
        else:
            if r in _AVOID:
                names = _AVOID[r]
            else:       # (HRNet's branch streams: beside main and beside each other)
                names = ("main",) + tuple(k for k in b if k.startswith("branch"))
            avoid = [_main(device) if n == "main" else b[n] for n in names if n == "main" or n in b]
            b[r] = _pick(device, avoid)
        if r in ("wgrad", "tokens", "tokens_wgrad") or r.startswith("branch"):
            from .dp import register_producer      # gradient kernels run there: collectives are ordered after it
            register_producer(b[r])
    return b


def get(device, role):
    """The stream of a role.  Roles of the train step that are not bound yet are bound in ORDER up to this one, so a
    model used without a trainer gets the same queues as one used with it."""
    b = bound(device)
    if role not in b:
        if role in ORDER:      # (the collective stream is bound only when somebody asks for it: the data-parallel layer)
            bind(device, [r for r in ORDER[:ORDER.index(role) + 1] if r == role or r != "comm"])
        else:
            bind(device, [role])
    return b[role]


def plan_for_collectives(device):
    """Call BEFORE the process group is created (scat_amd.dp.init_distributed does).  torch.distributed runs every
    collective on a stream ProcessGroupNCCL takes from torch's pool (32 streams per device and priority, handed out
    round-robin) when the communicator is created: the pool is walked so that the request ProcessGroupNCCL is about to
    make returns a stream that shares a queue with none of main / wgrad / tokens, and the stream the collectives are
    ordered after is picked the same way; the early optimiser update runs on that one (it waits for the collectives
    anyway).  RCCL's internal streams take what is left: they are idle once the communicator is up."""
    device = _dev(device)
    b = bound(device)
    if "_collective" in b:
        return
    bind(device, ["wgrad", "tokens"])
    heavy = [_main(device), b["wgrad"], b["tokens"]]
    bind(device, ["comm", "tokens_wgrad", "aux"])
    ring = [torch.cuda.Stream(device=device) for _ in range(32)]          # the 33rd request returns ring[0] again
    j = next((i for i, s in enumerate(ring) if not any(_shares(s, h) for h in heavy)), 0)
    for _ in range(j):                                                     # ... and after j more, ring[j]
        torch.cuda.Stream(device=device)
    b["_collective"] = ring[j]
    b["opt"] = b["comm"]
    _PLAN[device.index] = {"ring": ring, "j": j, "verified": None}


_PLAN = {}


def verify_collective_plan(device):
    """Call right AFTER the process group exists (scat_amd.dp.init_distributed does).  plan_for_collectives() relies on
    two things torch does not document: its stream pool is handed out round-robin, and ProcessGroupNCCL draws exactly
    one stream from it when its communicator comes up.  Both are CHECKED here instead of trusted: the next pool stream
    must be the successor of the one the plan left for the collectives (else something else drew from the pool, or the
    communicator was not created eagerly); and the heavy roles of the step — main, wgrad, tokens, comm and the
    collective stream — are measured pairwise once more; a role that shares a queue with another is re-picked (nothing
    has captured these streams yet at this point) and, if that does not help, named in a loud warning.
    -> {"pool_walk_ok": bool | None, "shared": [pairs still sharing], "repicked": [roles]}; also kept for bench.py."""
    import warnings

    device = _dev(device)
    b = bound(device)
    plan = _PLAN.get(device.index)
    res = {"pool_walk_ok": None, "shared": [], "repicked": []}
    if plan is not None:
        nxt = torch.cuda.Stream(device=device)
        want = plan["ring"][(plan["j"] + 1) % 32]
        res["pool_walk_ok"] = bool(nxt.cuda_stream == want.cuda_stream)
        if not res["pool_walk_ok"]:
            warnings.warn("scat_amd.streams: the process group did not take the pool stream the queue plan left for it "
                          "(torch's stream pool is not where plan_for_collectives() expected it): the collectives may "
                          "share a hardware queue with the train step — see config.hw_queues in the bench line",
                          RuntimeWarning)
    heavy = [n for n in ("wgrad", "tokens", "comm", "_collective") if n in b]

    def clashes():
        names = ["main"] + heavy
        sts = [_main(device)] + [b[n] for n in heavy]
        return [(names[i], names[k]) for i in range(len(sts)) for k in range(i + 1, len(sts))
                if _shares(sts[i], sts[k])]

    bad = clashes()
    for _ in range(2):
        movable = sorted({y for _, y in bad if y != "_collective"} | {x for x, y in bad if y == "_collective" and x != "main"})
        if not movable:
            break
        from . import dp
        for r in movable:
            old = b.pop(r)
            dp._PRODUCERS[:] = [s for s in dp._PRODUCERS if s is not old]
            if b.get("opt") is old:
                b.pop("opt")
            names = tuple(n for n in ("main", "wgrad", "tokens", "comm", "_collective") if n != r)
            b[r] = _pick(device, [_main(device) if n == "main" else b[n] for n in names if n == "main" or n in b])
            if r in ("wgrad", "tokens"):
                dp.register_producer(b[r])
            if r == "comm":
                b["opt"] = b[r]
            res["repicked"].append(r)
        bad = clashes()
    res["shared"] = ["+".join(p) for p in bad]
    if bad:
        warnings.warn(f"scat_amd.streams: {res['shared']} share a hardware queue after re-picking: the collectives (or the "
                      "weight gradients) will serialise behind the other stream's kernels (~4 ms of a 23 ms step on "
                      "MI355X, profiles/r03_dp_queues.txt)", RuntimeWarning)
    if plan is not None:
        plan["verified"] = res
    return res


def plan_report(device):
    """what verify_collective_plan() found on this rank (None: no process group was planned)"""
    plan = _PLAN.get(_dev(device).index)
    return None if plan is None else plan["verified"]


def alias(device, role, to):
    """``role`` runs on the stream of role ``to`` from now on (the early optimiser update on the collective stream)."""
    b = bound(device)
    b[role] = get(device, to)
    return b[role]


def sharing(device):
    """Diagnostic: the pairs among the bound roles (and the null stream, 'main') that share a hardware queue."""
    device = _dev(device)
    b = dict(bound(device))
    names = ["main"] + list(b)
    sts = [_main(device)] + [b[n] for n in names[1:]]
    out = []
    for i in range(len(sts)):
        for j in range(i + 1, len(sts)):
            if sts[i] is sts[j] or sts[i].cuda_stream == sts[j].cuda_stream:
                out.append((names[i], names[j], "same stream"))
            elif _shares(sts[i], sts[j]):
                out.append((names[i], names[j], "same queue"))
    return out
