"""A whole train step (forward + backward + update) captured as ONE HIP graph — a measurement facility, not the product path.

Question it answers: HRNet-W32's step is ~2 400 kernels of 10-50 us issued by ~1 800 library calls, and the host needs 47-57 ms
to issue what the device finishes in 55-60 (tools/cpu_issue.py, tools/host_profile.py) — would replaying a captured step,
whose host share is one hipGraphLaunch, be faster?  Measured (tools/graph_probe.py, profiles/r04_graph_probe*.txt, the
full multi-stream step of each network, same process): HRNet-W32 eager 59.4 ms, replay 60.9; ResNet-50 eager 22.7, replay 23.5.
**No**: the device is the bound in both, and this runtime's graph replay runs the same kernels slightly slower than the four
hardware queues fed by streams do.  The bench and the trainer therefore stay eager.  (A replay also freezes what the host
computes per step — Adam's bias correction, the masked-token draw — so ``GraphedStep`` as it stands is for timing only.)

What the experiment left behind is that the steps ARE capturable, and what that takes on this stack (ROCm 7.2; each rule was a
segmentation fault inside hipStreamEndCapture before it was a rule):

* **Star-shaped fork / join.**  Only the capturing stream may bring another stream into the capture, and side streams may
  wait on nothing but it: a side stream that waits on another side stream — in either direction, even when both were forked
  from the capturing stream first — crashes the end of the capture.  The library's side streams are forked from / joined to
  the stream they are called on; the two places where that stream is itself a side stream check ``fork_ok()`` (the token
  mixer's weight-gradient stream under the ResNet wrappers' token stream) or route through relay nodes on the caller's
  stream (HRNet's exchange units, models/hrnet.py ``_RelayFn`` — which also made the eager step 2-3 % faster).
* **No default stream.**  autograd's AccumulateGrad nodes remember the stream they were created on; an eager step on the
  legacy default stream leaves state that pulls it into the capture through a side stream.  ``GraphedStep`` runs every
  step — the warm-up ones included — on a stream of its own.
"""
from __future__ import annotations

import torch

_capturing_on = None      # the capturing stream's handle while a step is being captured


def capturing() -> bool:
    return _capturing_on is not None


def fork_ok() -> bool:
    """May the CURRENT stream fork a side stream?  Always, except while a step is being captured and the current stream is
    not the capturing one (see the module docstring: star-shaped fork / join)."""
    if _capturing_on is None:
        return True
    return torch.cuda.current_stream().cuda_stream == _capturing_on


class GraphedStep:
    """``body()`` (no arguments: it reads its inputs from tensors that stay where they are) run eagerly ``warmup`` times on a
    stream of its own, then captured; every later call replays.  ``before`` (optional) runs on the host before each call —
    eager, capture or replay — to refresh static inputs (uploads ordered on the step's stream).  Returns what ``body``
    returned at capture (tensors that each replay overwrites)."""

    def __init__(self, body, warmup=3, before=None):
        self.body, self.warmup, self.before = body, warmup, before
        self.calls = 0
        self.graph = None
        self.out = None
        self.stream = torch.cuda.Stream()

    def __call__(self):
        global _capturing_on
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            if self.before is not None:
                self.before()
            if self.graph is not None:
                self.graph.replay()
            elif self.calls < self.warmup:
                self.out = self.body()
            else:
                g = torch.cuda.CUDAGraph()
                _capturing_on = self.stream.cuda_stream
                try:
                    with torch.cuda.graph(g, stream=self.stream):
                        self.out = self.body()
                finally:
                    _capturing_on = None
                self.graph = g
                g.replay()          # capture only records: this is the step itself
        self.calls += 1
        cur.wait_stream(self.stream)
        return self.out
