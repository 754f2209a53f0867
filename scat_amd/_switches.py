"""A/B switches of the host side.

Every scheduling / fusion choice of the step that was decided by a same-box measurement keeps
its losing side reachable — for the numerical A/B test (tools/ab_check.py, tests/test_gpu_schedule.py) and for re-measuring
on new hardware — but only when ``SCAT_DIAG=1`` is set: without it the environment is not consulted at all and the product's
behaviour depends on its inputs alone.  ``REGISTRY`` lists every switch with its default, so the A/B test can flip all of
them and INTEGRATION.md can list them.
"""
from __future__ import annotations

import os

DIAG = os.environ.get("SCAT_DIAG", "0") != "0"
REGISTRY = {}


def ab(name: str, default: bool) -> bool:
    REGISTRY[name] = default
    if not DIAG:
        return default
    v = os.environ.get(name)
    return default if v is None else v != "0"


def ab_int(name: str, default: int) -> int:
    REGISTRY[name] = default
    if not DIAG:
        return default
    v = os.environ.get(name)
    return default if v is None else int(v)
