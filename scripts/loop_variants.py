"""Which call of execute() costs wall-clock beyond its own duration?  Loops of mik_factor + mik_predict on a bench config with
the other calls of an execute() added one at a time (ms per iteration, 6 iterations after 2 warm-ups)."""
import os, sys, time
os.environ["MIK_FACTOR_CACHE"] = "0"
import numpy as np
sys.path.insert(0, ".")
from bench import CONFIGS, synth, make_model, grid_axes

c = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = CONFIGS[c]
coords, values = synth(cfg["seed"], cfg["n"], cfg["ndim"])
m = make_model(cfg, coords, values)
axes = grid_axes(cfg, 1)
h = m._get_handle()
m._set_problem(h)
P = m._prepare("grid", axes, None)
P.load(h, cfg["ndim"])

def loop(name, body, n=6):
    for _ in range(2):
        body()
    h.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        body()
    h.synchronize()
    print("%-58s %.3f ms per iteration" % (name, (time.perf_counter() - t0) / n * 1e3), flush=True)

def a():
    h.factor(); h.predict()
def b():
    m._set_problem(h); h.factor(); h.predict()
def c_():
    m._set_problem(h); h.factor(); P.load(h, cfg["ndim"]); h.predict()
keep = [None]
def d():
    m._set_problem(h); h.factor(); P.load(h, cfg["ndim"]); h.predict(); keep[0] = h.get_results()
def e():
    h.factor(); h.predict(); keep[0] = h.get_results()
def f():
    h.factor(); h.predict(); time.sleep(0.0005)
loop("factor + predict (the resident step)", a)
loop("set_problem + factor + predict", b)
loop("set_problem + factor + set_grid + predict", c_)
loop("set_problem + factor + set_grid + predict + get_results", d)
loop("factor + predict + get_results (zero-copy take)", e)
loop("factor + predict + 0.5 ms of host sleep", f)
loop("factor + predict (again)", a)
loop("execute('grid') of the class", lambda: keep.__setitem__(0, m.execute("grid", *axes, backend="loop")))
