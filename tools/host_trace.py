"""Host-side launch timeline of one training step (headline shape): wraps every specforge_amd.ops function and the
library's check() to time-stamp the host at each launch; prints the largest host-side gaps between consecutive launches
of the LAST of 4 steps, so a blocking call / allocation / GC pause in the launch path shows up by name.   (GPU box)"""
import gc
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from specforge_amd import ops  # noqa: E402

events = []


def wrap(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        events.append((t0, time.perf_counter(), name))
        return r
    return w


for n in dir(ops):
    f = getattr(ops, n)
    if callable(f) and not n.startswith("_") and getattr(f, "__module__", "") == ops.__name__:
        setattr(ops, n, wrap(n, f))

import collections
import traceback

sync_sites = collections.Counter()


def spy(cls, name):
    orig = getattr(cls, name)

    def w(self, *a, **k):
        t0 = time.perf_counter()
        r = orig(self, *a, **k)
        dt = time.perf_counter() - t0
        if dt > 2e-3:   # a call that blocked the host for > 2 ms: remember who made it
            fr = [f"{f.filename.split('/')[-1]}:{f.lineno}" for f in traceback.extract_stack(limit=7)[:-1]]
            sync_sites[(name, " < ".join(reversed(fr)))] += 1
        return r
    setattr(cls, name, w)


for nm in ("item", "tolist", "cpu", "__bool__", "__float__", "__int__", "numpy"):
    spy(torch.Tensor, nm)
spy(torch.cuda.Event, "synchronize")

from specforge_amd import eagle3 as _e3, engine as _eng, training as _tr  # noqa: E402
marks = []


def mark_wrap(cls, name, label):
    orig = getattr(cls, name)

    def w(*a, **k):
        marks.append((time.perf_counter(), label + ":enter"))
        r = orig(*a, **k)
        marks.append((time.perf_counter(), label + ":exit"))
        return r
    setattr(cls, name, staticmethod(w) if isinstance(cls.__dict__.get(name), staticmethod) else w)


mark_wrap(_e3.Eagle3TrainStrategy, "forward_loss", "forward_loss")
mark_wrap(_tr.HipDPTrainingBackend, "backward", "backend.backward")
mark_wrap(_tr.HipDPTrainingBackend, "step", "backend.step")
mark_wrap(_eng.Eagle3Engine, "backward", "engine.backward")
mark_wrap(_eng.Eagle3Engine, "forward", "engine.forward")

sys.argv = ["bench.py", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-dense-mask"]
if "--nogc" in sys.argv or True:
    pass
t_mark = []
orig_sync = torch.cuda.synchronize


def sync(*a, **k):
    t_mark.append(time.perf_counter())
    return orig_sync(*a, **k)


torch.cuda.synchronize = sync
bench.main()
# last step = launches after the second-to-last adamw
idx = [i for i, e in enumerate(events) if e[2] == "adamw_step"]
ev = events[idx[-2] + 1: idx[-1] + 1]
print("launches in last step:", len(ev), " host span ms:", round((ev[-1][1] - ev[0][0]) * 1e3, 2))
gaps = sorted(((b[0] - a[1]) * 1e3, a[2], b[2], i) for i, (a, b) in enumerate(zip(ev, ev[1:])))[-8:]
for g in reversed(gaps):
    print(f"{g[0]:8.3f} ms host gap after {g[1]} -> before {g[2]} (launch #{g[3]})")
inside = sorted(((e[1] - e[0]) * 1e3, e[2]) for e in ev)[-5:]
print("longest calls:", [(round(a, 3), b) for a, b in reversed(inside)])
print("gc counts", gc.get_count(), "thresholds", gc.get_threshold())

for (name, site), n in sync_sites.most_common(8):
    print(n, "x blocking", name, "at", site)

last = [m for m in marks if m[0] >= ev[0][0] - 0.05 and m[0] <= ev[-1][1] + 0.01]
for (t, l), (t2, l2) in zip(last, last[1:]):
    print(f"{(t2 - t) * 1e3:9.3f} ms  {l} -> {l2}")
