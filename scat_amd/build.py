"""Build libscat_hip.so (gfx950) in-tree with hipcc. No torch dependency in the library."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libscat_hip.so")
OBJ = os.path.join(HERE, "csrc", "_obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: hipcc's SLP vectoriser turns adjacent scalar fp32 adds / multiplies into packed fp32 instructions
# (v_pk_add_f32, v_pk_fma_f32), which cost ~17 issue cycles against 4 beside a busy matrix pipe (MI355X_MICROARCH.md,
# "price of one filler beside MFMAs ... an anti-lever") — every operand split is such a pair.  Measured on the whole
# library: weight gradients 3.95 -> 3.67 ms (tools/conv_bench.py), train step 23.66 -> 23.22 ms, same box.
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    m = 0.0
    for root, _, files in os.walk(CSRC):
        if root.endswith(("_obj", "_obj_diag")):
            continue
        for f in files:
            if f.endswith((".h", ".hip")):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    m = max(m, os.path.getmtime(os.path.join(HERE, "..", "include", "scat_hip.h")))
    return m


DIAG_OUT = os.path.join(HERE, "..", "tools", "_bin", "libscat_hip_diag.so")


def build(force: bool = False, verbose: bool = True, diag: bool = False) -> str:
    """Compile every .hip for gfx950 and link the shared library. Returns its path.
    diag: the tools build (-DSCAT_DIAG: in-kernel time stamps, ablation and negative-result kernel variants) into
    tools/_bin/libscat_hip_diag.so — never the library the boundary ships; tools select it with SCAT_LIBPATH."""
    global OUT, OBJ
    if diag:
        out, obj, flags = os.path.abspath(DIAG_OUT), os.path.join(CSRC, "_obj_diag"), FLAGS + ["-DSCAT_DIAG"]
        os.makedirs(os.path.dirname(out), exist_ok=True)
        saved = (OUT, OBJ, list(FLAGS))
        OUT, OBJ = out, obj
        FLAGS[:] = flags
        try:
            return build(force, verbose, False)
        finally:
            OUT, OBJ = saved[0], saved[1]
            FLAGS[:] = saved[2]
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= _deps_mtime():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    hdr_m = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h"))
    hdr_m = max(hdr_m, os.path.getmtime(os.path.join(HERE, "..", "include", "scat_hip.h")))

    def one(src):
        o = os.path.join(OBJ, src[:-4] + ".o")
        s = os.path.join(CSRC, src)
        if not force and os.path.exists(o) and os.path.getmtime(o) >= max(os.path.getmtime(s), hdr_m):
            return o
        cmd = [HIPCC, *FLAGS, "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return o

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(one, _sources()))
    cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", OUT, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, diag="--diag" in sys.argv))
