"""Numerical A/B of the scheduling switches: a few train steps at an odd batch size with every overlap / fold / cache
enabled versus all of them disabled (single stream, per-call weight re-layout, unfolded BatchNorm backward, late
Adam, BatchNorm statistics from a pass over the convolution output instead of its epilogue).  The two runs must give the same losses up to fp32 summation-order effects."""
import os
import subprocess
import sys

# every boolean switch of scat_amd._switches.REGISTRY that the ResNet-50 step consults, flipped to its non-default side
# (honoured only with SCAT_DIAG=1); the HRNet-only switches do not touch this network
OFF = dict(SCAT_DIAG="1", SCAT_OVERLAP_TOKENS="0", SCAT_SIDE_WGRAD="0", SCAT_EARLY_ADAM="0", SCAT_WPREP="0", SCAT_BNB="0",
           SCAT_SUBSAMPLE="0", SCAT_DX2_FOLD="0", SCAT_SIDE_HEAD="0", SCAT_EPI_STATS="0", SCAT_SIDE_SHORTCUT="0",
           SCAT_SIDE_DS_BN="0", SCAT_STEM_FUSED_BWD="0", SCAT_STAT_REF="0", SCAT_GROUP_WGRAD="0", SCAT_TOKENS_FIRST="0",
           SCAT_STREAMS_PROBE="0", SCAT_BN_ONEPASS="0", SCAT_BN_LASTBLOCK="0", SCAT_EPI_BNB="0")
CHILD = r'''
import sys, random, torch
sys.path.insert(0, ".")
import bench
from scat_amd.trainer import TrainStep
B = int(sys.argv[1])
dev = torch.device("cuda", 0)
net = bench.make_net("resnet50", 1, dev)
ts = TrainStep(net, lr=1e-4)
u8, lab = bench.build_inputs(B, 100, dev)
from scat_amd import ops as _ops
x = _ops.preprocess_u8(u8, (224, 224))
random.seed(3)
out = []
for i in range(4):
    out.append(float(ts(x, lab)[0]))
print("LOSSES", " ".join(repr(v) for v in out))
'''


def run(batch, extra):
    env = dict(os.environ, **extra)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", CHILD, str(batch)], env=env, capture_output=True, text=True, timeout=600,
                       cwd=root)
    line = [l for l in r.stdout.splitlines() if l.startswith("LOSSES")]
    if not line:
        raise RuntimeError(r.stdout[-2000:] + r.stderr[-2000:])
    return [float(v) for v in line[0].split()[1:]]


if __name__ == "__main__":
    for batch in (13, 100):
        a, b = run(batch, {}), run(batch, OFF)
        rel = [abs(u - v) / abs(v) for u, v in zip(a, b)]
        print(f"batch {batch}: all on {a}\n           all off {b}\n           rel diff {['%.1e' % r for r in rel]}")
        assert rel[0] < 1e-5 and max(rel) < 5e-2, rel
    print("ok")
