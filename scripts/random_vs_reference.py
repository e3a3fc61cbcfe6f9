"""Randomized differential run on the GPU box against the REAL reference (oracle/_ref), not the oracle restatement: random class, variogram (given or
fitted), anisotropy, drifts, exactness, pseudo-inverse, coordinate dtype, style, backend, window -- one constructor + one execute() per case, both sides
fed the same objects.  Both must return the same (|dz| <= 1e-8, |dsigma^2| <= 1e-6 scaled by the magnitudes involved, shape, masked-array-ness, mask) or
raise the same kind.  `python scripts/random_vs_reference.py [N] [seed] [-v]` (-v: also report the ill-conditioned cases that are left out; --near-origin: no 1e5 coordinate offsets, which make most regional_linear systems ill-conditioned upstream); exits non-zero on a disagreement."""
import sys
import warnings

import numpy as np

sys.path.insert(0, ".")
import pykrige_amd as pa  # noqa: E402
from oracle import ref_package as rp  # noqa: E402

pk = rp.import_reference(stub_statistics=True)
ARGS = [a for a in sys.argv[1:] if not a.startswith("-")]
N = int(ARGS[0]) if ARGS else 300
SEED = int(ARGS[1]) if len(ARGS) > 1 else 606
MODELS = ["linear", "power", "gaussian", "spherical", "exponential", "hole-effect"]


def params_for(model, r):
    if model == "linear":
        return [float(r.uniform(0.3, 2.0)), float(r.uniform(0.0, 0.2))]
    if model == "power":
        return [float(r.uniform(0.3, 2.0)), float(r.uniform(0.3, 1.8)), float(r.uniform(0.0, 0.2))]
    return [float(r.uniform(0.5, 2.0)), float(r.uniform(0.2, 0.9)), float(r.uniform(0.0, 0.2))]


def case(i):
    r = np.random.default_rng([SEED, i])
    dim3 = r.random() < 0.3
    universal = r.random() < 0.45
    n = int(r.integers(8, 140))
    xs = [r.random(n) * r.choice([1.0, 10.0, 1000.0]) + r.choice([0.0, -5.0] if "--near-origin" in sys.argv else [0.0, -5.0, 1e5]) for _ in range(3 if dim3 else 2)]
    span = [a.max() - a.min() for a in xs]
    v = np.sin(3 * (xs[0] - xs[0].min()) / span[0]) + (xs[1] - xs[1].min()) / span[1] + 0.1 * r.standard_normal(n)
    model = str(r.choice(MODELS))
    kw = {"variogram_model": model}
    geographic = (not dim3) and (not universal) and r.random() < 0.2
    if geographic:  # lon / lat in degrees, ranges in degrees of arc; no anisotropy on the sphere
        xs = [r.uniform(-180, 180, n) if r.random() < 0.5 else r.uniform(150, 210, n) % 360, r.uniform(-70, 70, n)]
        span = [40.0, 40.0]
        v = np.sin(np.radians(xs[0])) + np.cos(np.radians(xs[1])) + 0.1 * r.standard_normal(n)
        kw["coordinates_type"] = "geographic"
    custom = r.random() < 0.08
    if custom:
        model = "custom"
        kw["variogram_model"] = "custom"
        c0, c1, c2 = float(r.uniform(0.5, 2)), float(np.mean(span) * r.uniform(0.2, 0.8)), float(r.uniform(0.0, 0.2))
        kw["variogram_parameters"] = [c0, c1, c2]
        kw["variogram_function"] = lambda p, d: p[0] * (1.0 - np.exp(-d / p[1])) + p[2]
    if custom:
        pass
    elif r.random() < 0.8:
        p = params_for(model, r)
        if model not in ("linear", "power"):
            p[1] *= float(np.mean(span))  # a range in the units of the coordinates
        elif model == "linear":
            p[0] /= float(np.mean(span))
        kw["variogram_parameters"] = p if r.random() < 0.5 else dict(zip({"linear": ["slope", "nugget"], "power": ["scale", "exponent", "nugget"]}.get(model, ["sill", "range", "nugget"]), p))
    else:
        kw["nlags"] = int(r.integers(4, 9))
        kw["weight"] = bool(r.random() < 0.5)
    kw["exact_values"] = bool(r.random() < 0.8)
    if r.random() < 0.15:
        kw["pseudo_inv"] = True
    if dim3:
        if r.random() < 0.5:
            kw.update(anisotropy_scaling_y=float(r.uniform(0.5, 3)), anisotropy_scaling_z=float(r.uniform(0.5, 3)), anisotropy_angle_x=float(r.uniform(0, 90)),
                      anisotropy_angle_y=float(r.uniform(0, 90)), anisotropy_angle_z=float(r.uniform(0, 90)))
    elif r.random() < 0.5 and not geographic:
        kw.update(anisotropy_scaling=float(r.uniform(0.5, 3)), anisotropy_angle=float(r.uniform(0, 180)))
    drift_note = ""
    if universal:
        terms = []
        if r.random() < 0.6:
            terms.append("regional_linear")
        if not dim3 and r.random() < 0.3:
            terms.append("point_log")
            kw["point_drift"] = np.column_stack([xs[0].min() + span[0] * r.random(2), xs[1].min() + span[1] * r.random(2), r.uniform(-1, 1, 2)])
        if r.random() < 0.25:
            terms.append("functional")
            # (bounded and NOT a linear function of the coordinates: beside regional_linear a linear one makes the matrix singular)
            x0, s0, y0, s1 = float(xs[0].min()), float(span[0]), float(xs[1].min()), float(span[1])
            kw["functional_drift"] = [(lambda a, b, c: np.sin(2.0 * (a - x0) / s0) * np.cos((b - y0) / s1)) if dim3 else (lambda a, b: np.sin(2.0 * (a - x0) / s0) * np.cos((b - y0) / s1))]
        spec = r.random() < 0.2
        if spec:
            terms.append("specified")
            kw["specified_drift"] = [np.cos(3.0 * (xs[0] - xs[0].min()) / span[0])]
        ext = (not dim3) and r.random() < 0.2
        if ext:
            terms.append("external_Z")
            ex = np.linspace(xs[0].min() - 0.2 * span[0], xs[0].max() + 0.2 * span[0], int(r.integers(4, 12)))
            ey = np.linspace(xs[1].min() - 0.2 * span[1], xs[1].max() + 0.2 * span[1], int(r.integers(4, 12)))
            kw.update(external_drift=r.random((ey.size, ex.size)), external_drift_x=ex, external_drift_y=ey)
        kw["drift_terms"] = terms
        drift_note = "+".join(terms) or "no terms"
    dt = r.choice([np.float64, np.float64, np.float32])
    style = str(r.choice(["grid", "points", "masked"]))
    backend = str(r.choice(["vectorized", "loop"] if (universal or dim3) else ["vectorized", "loop", "C"]))
    axes = [np.linspace(a.min() - 0.05 * s, a.max() + 0.05 * s, int(r.integers(2, 9))).astype(dt) for a, s in zip(xs, span)]
    if geographic:
        dt = np.float64  # (float32 lon / lat: upstream's great-circle arithmetic runs in float32 on the point side -- reported by edge_forms_vs_reference.py, not restated)
        axes = [np.linspace(xs[0].min(), xs[0].max(), int(r.integers(2, 9))), np.linspace(-60, 60, int(r.integers(2, 9)))]
    ekw = {"backend": backend}
    if universal and "specified" in kw.get("drift_terms", ()):
        backend = "vectorized"  # (the loop backend indexes the specified drift by the matrix row upstream: uk.py:1070)
        ekw["backend"] = backend
    if style == "points":
        m = int(r.integers(1, 30))
        axes = [(a.min() + s * r.random(m)).astype(dt) for a, s in zip(xs, span)]
        if r.random() < 0.3:  # some points ON stations
            k = min(m, 3)
            for ax, a in zip(axes, xs):
                ax[:k] = a[:k].astype(dt)
    if style == "masked":
        ekw["mask"] = r.random(tuple(a.size for a in reversed(axes))) < 0.4
    if universal and "specified" in kw.get("drift_terms", ()):
        if style == "points":
            ekw["specified_drift_arrays"] = [np.cos(3.0 * (axes[0].astype(np.float64) - xs[0].min()) / span[0])]
        else:
            g = np.cos(3.0 * (axes[0].astype(np.float64) - xs[0].min()) / span[0])
            ekw["specified_drift_arrays"] = [np.broadcast_to(g, tuple(a.size for a in reversed(axes))).copy()]
    if not universal and backend != "vectorized" and r.random() < 0.35:
        ekw["n_closest_points"] = int(r.integers(2, min(n, 20) + 1))
    name = "%s%s %s n=%d %s %s[%s]%s %s%s" % ("UK" if universal else "OK", "3D" if dim3 else "2D", model, n, "given" if "variogram_parameters" in kw else "fitted", style, backend,
                                           " k=%d" % ekw["n_closest_points"] if "n_closest_points" in ekw else "", np.dtype(dt).name, (" " + drift_note if universal else "") + (" geographic" if geographic else ""))

    def make(mod):
        cls = {(False, False): mod.ok.OrdinaryKriging, (True, False): mod.uk.UniversalKriging, (False, True): mod.ok3d.OrdinaryKriging3D, (True, True): mod.uk3d.UniversalKriging3D}[(universal, dim3)]
        return cls(*xs, v, **kw)

    return name, make, (lambda m: m.execute(style, *axes, **ekw)), float(np.abs(v).max())


def main():
    bad, agree, raised, noted, illcond = [], 0, 0, 0, 0
    worst = [0.0, 0.0]
    for i in range(N):
        name, make, call, scale = case(i)
        out = []
        for mod in (pk, pa):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                try:
                    out.append(("ok", call(make(mod))))
                except Exception as e:  # noqa: BLE001
                    out.append(("raise", e))
        (ka, ra), (kb, rb) = out
        if ka == "raise" and isinstance(ra, NotImplementedError) and kb == "ok":
            noted += 1  # backend='C' knows five models (lib/variogram_models.pyx:9-20): a deliberate deviation, the drop-in kriges
            continue
        cond = None
        if kb == "ok":
            try:  # the conditioning of the system both solved (the drop-in's assembled matrix): beyond 1e9 the reference's own digits are gone
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    cond = float(np.linalg.cond(make(pa)._get_kriging_matrix()))
            except Exception:  # noqa: BLE001
                cond = None
        if cond is not None and not cond < 1e9:
            illcond += 1
            if ka == kb == "ok" and "-v" in sys.argv:  # reported only: what the two made of a system whose digits are gone
                dz = float(np.nanmax(np.abs(np.ma.filled(np.ma.asarray(ra[0]), 0.0) - np.ma.filled(np.ma.asarray(rb[0]), 0.0)))) if np.size(ra[0]) else 0.0
                print("%4d %-90s ill-conditioned: cond_2 %.1e, max|dz| %.1e (cond x 1e-16 x |z| = %.1e)" % (i, name, cond, dz, cond * 1e-16 * scale))
            elif ka != kb and "-v" in sys.argv:
                print("%4d %-90s ill-conditioned: cond_2 %.1e, reference %s, drop-in %s" % (i, name, cond, ka if ka == "ok" else type(ra).__name__, kb if kb == "ok" else type(rb).__name__))
            continue
        if ka != kb:
            bad.append(name)
            print("%4d %-90s DISAGREE: reference %s, drop-in %s" % (i, name, ka if ka == "ok" else type(ra).__name__ + ": " + str(ra)[:60], kb if kb == "ok" else type(rb).__name__ + ": " + str(rb)[:80]))
            continue
        if ka == "raise":
            raised += 1
            if not (isinstance(rb, type(ra)) or isinstance(ra, type(rb))):
                bad.append(name)
                print("%4d %-90s raise DIFFERENT kinds: %s / %s" % (i, name, type(ra).__name__, type(rb).__name__))
            continue
        msg = []
        for j, (a, b, tol) in enumerate(((ra[0], rb[0], 1e-8), (ra[1], rb[1], 1e-6))):
            if np.shape(a) != np.shape(b) or np.ma.isMaskedArray(a) != np.ma.isMaskedArray(b) or not np.array_equal(np.ma.getmaskarray(a), np.ma.getmaskarray(b)):
                msg.append("shape / mask of output %d" % j)
                continue
            da, db = np.ma.filled(np.ma.asarray(a), 0.0), np.ma.filled(np.ma.asarray(b), 0.0)
            if da.size:
                d = np.abs(da - db)
                fin = np.isfinite(da) & np.isfinite(db)
                if not (np.isfinite(da) == np.isfinite(db)).all():
                    msg.append("non-finite pattern of output %d" % j)
                m = float(d[fin].max()) if fin.any() else 0.0
                worst[j] = max(worst[j], m / max(1.0, scale if j == 0 else 1.0))
                if m > tol * max(1.0, scale if j == 0 else 1.0, float(np.abs(da[fin]).max()) if fin.any() else 1.0):
                    msg.append("output %d max|d| %.2e (values up to %.2e)" % (j, m, float(np.abs(da[fin]).max())))
        if msg:
            bad.append(name)
            print("%4d %-90s DISAGREE: %s" % (i, name, "; ".join(msg)))
        else:
            agree += 1
    print("%d cases (seed %d): %d agree in value, %d raise alike, %d disagree; %d left out as ill-conditioned (cond_2 >= 1e9), %d where backend='C' does not know the model upstream; "
          "worst |dz| %.2e, |dsigma^2| %.2e" % (N, SEED, agree, raised, len(bad), illcond, noted, worst[0], worst[1]))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
