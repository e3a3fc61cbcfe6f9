"""diagnostic (round 6): the bench's k = 10 line takes 5.1 ms per execute() inside the default run and 2.1 ms on its own -- which call grows, and after what?"""
import collections
import os
import sys
import time
sys.path.insert(0, ".")
os.environ["MIK_FACTOR_CACHE"] = "0"
import bench
from pykrige_amd import _lib

acc = collections.OrderedDict()


def wrap(name):
    f = getattr(_lib.Handle, name)

    def g(self, *a, **k):
        t0 = time.perf_counter()
        try:
            return f(self, *a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    setattr(_lib.Handle, name, g)


for n in ("set_problem", "set_grid", "predict_moving_window", "timing", "get_results", "synchronize", "close"):
    wrap(n)


def line(tag):
    acc.clear()
    l = bench.other_config_line(2, 10, steps=3, warmup=1)
    print("%-46s k = 10: %.3f ms per step; Handle calls (sum over 1 + 3 + parity calls, ms): %s" % (tag, l["ms_per_step"], {k: round(v * 1e3, 2) for k, v in acc.items()}), flush=True)


line("fresh process")
line("again")
for cno in (3, 4, 5):
    bench.other_config_line(cno, None, steps=1, warmup=1)
    line("after config %d" % cno)
# where the time of a step goes in that state: cProfile over 8 more calls
import cProfile, io, pstats
cfg2 = bench.CONFIGS[2]
co, va = bench.synth(cfg2["seed"], cfg2["n"], 2)
mm = bench.make_model(cfg2, co, va)
ax = bench.grid_axes(cfg2, 1)
for _ in range(2):
    mm.execute("grid", *ax, backend="loop", n_closest_points=10)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(8):
    mm.execute("grid", *ax, backend="loop", n_closest_points=10)
pr.disable()
print("8 calls: %.3f ms per call" % ((time.perf_counter() - t0) / 8 * 1e3))
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(12)
print("\n".join(st.getvalue().splitlines()[4:26]))
cfg = bench.CONFIGS[2]
coords, values = bench.synth(cfg["seed"], cfg["n"], 2)
m = bench.make_model(cfg, coords, values)
axes = bench.grid_axes(cfg, 1)
res = m.execute("grid", *axes, backend="loop")
line("after a config-2 execute, its handle + result alive")
parts = bench.shard_points(cfg, 0, 1)
zz = m.execute("points", *parts, backend="loop")
line("after a points-style execute")
