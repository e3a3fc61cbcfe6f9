import sys, time, gc, subprocess, numpy as np
sys.path.insert(0, ".")
import pykrige_amd as pa
rng = np.random.default_rng(1)
def used():
    out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showmeminfo", "vram", "--csv"], capture_output=True, text=True).stdout
    try: return int(out.strip().splitlines()[-1].split(",")[2]) / 2**20
    except Exception: return float("nan")
gx = np.linspace(0, 1, 64)
print("start: %.0f MiB" % used())
t0 = time.perf_counter(); keep = []
for i in range(300):
    n = 500
    x, y, v = rng.random(n), rng.random(n), rng.random(n)
    m = pa.OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1., .3, .05])
    z, s = m.execute("grid", gx, gx)
    if i % 100 == 99:
        gc.collect(); print("after %d short-lived objects: %.0f MiB, %.1f ms per construct+execute" % (i + 1, used(), (time.perf_counter() - t0) / (i + 1) * 1e3))
for i in range(40):
    n = 3000
    x, y, v = rng.random(n), rng.random(n), rng.random(n)
    m = pa.OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1., .3, .05])
    m.execute("grid", gx, gx); keep.append(m)
print("40 live objects of N = 3000: %.0f MiB" % used())
a, b = keep[0], keep[1]
za, _ = a.execute("grid", gx, gx); zb, _ = b.execute("grid", gx, gx); za2, _ = a.execute("grid", gx, gx)
print("interleaved objects reproduce:", np.array_equal(za, za2), "and differ from each other:", not np.array_equal(za, zb))
del keep, a, b, m; gc.collect(); print("after dropping them: %.0f MiB" % used())
import threading
ms = [pa.OrdinaryKriging(rng.random(800), rng.random(800), rng.random(800), variogram_model="spherical", variogram_parameters=[1., .3, .05]) for _ in range(4)]
ref = [mm.execute("grid", gx, gx)[0].copy() for mm in ms]
out = [None] * 4
def work(k):
    for _ in range(20): out[k] = ms[k].execute("grid", gx, gx)[0].copy()
th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
[t.start() for t in th]; [t.join() for t in th]
print("four objects on four threads reproduce their serial results:", all(np.array_equal(r, o) for r, o in zip(ref, out)))
