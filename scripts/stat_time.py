import time, numpy as np, sys
sys.path.insert(0, '.')
import pykrige_amd as pa
for n in (1000, 2000, 5000):
    rng = np.random.default_rng(2); x, y = rng.random(n), rng.random(n); v = np.sin(6*x)*np.cos(4*y)+0.1*rng.standard_normal(n)
    ok = pa.OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.0])
    ok._compute_statistics(); t=time.perf_counter(); ok._compute_statistics(); print(n, "statistics: %.3f s"%(time.perf_counter()-t), ok.get_statistics())
