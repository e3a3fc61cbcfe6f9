"""Host-side mirror of the reference's kriging classes for the execute() path.

Same class names, constructor keywords, ``execute`` signatures, return shapes and exceptions as
PyKrige 1.7.3 (/root/reference/src/pykrige: ok.py:187-206,760-1020; uk.py:220-244,1090-1328;
ok3d.py:198-219,735-932; uk3d.py:215-239,877-1146).  Everything between
``a = self._get_kriging_matrix(n)`` and ``return zvalues, sigmasq`` runs on the MI355X through
libmikrige.so (include/mikrige.h); this module only does the front/back matter the reference does in
Python: argument checks, meshgrid order, mask transposition, anisotropy adjustment, host-evaluated
drift terms, output shaping.

``backend``: the reference's values 'vectorized', 'loop' (and 'C' for OrdinaryKriging) are accepted
and all run the same device path -- they keep only their return-type convention ('vectorized' returns
MaskedArrays in every style, the others plain ndarrays unless style='masked').  'hip' is an explicit
alias with the 'loop'/'C' convention.  There is no CPU path in this package.
"""
import os as _os
import warnings

import numpy as np

from . import _lib
from . import core
from . import variogram_models


class _Cols:
    """The coordinate arrays of style='points' as they came (float64, one array per axis), behind the two-dimensional indexing the
    rest of this module uses on an (n, d) point array: cols[:, k], cols[rows, k], cols[slice], cols.shape.  Nothing is copied: the
    library reads the caller's arrays (it stages them in page-locked memory itself) -- the reference's np.array(..., copy=True)
    (ok.py:872-873) protects arrays it goes on to modify; these are only read."""

    __slots__ = ("cols", "shape")

    def __init__(self, cols):
        self.cols = []
        for c in cols:  # read-only VIEWS of the caller's arrays: an accidental in-place write raises instead of reaching user data
            c = c.view()
            c.flags.writeable = False
            self.cols.append(c)
        self.shape = (self.cols[0].shape[0], len(self.cols))

    def __getitem__(self, key):
        if isinstance(key, tuple):
            rows, k = key
            return self.cols[k][rows]
        return _Cols([c[key] for c in self.cols])

    def __array__(self, dtype=None, copy=None):  # np.asarray(cols): the (n, d) array (a copy; nothing on the execute path asks for it)
        if copy is False:  # NumPy 2 protocol: a caller that forbids copying cannot be served (the columns are separate arrays)
            raise ValueError("the (n, d) array of a point list is always a copy")
        a = np.stack(self.cols, axis=1)
        return a if dtype is None else a.astype(dtype, copy=False)


class _Pts:
    """The prediction points of one execute(): adjusted coordinate arrays (style='points', or a grid whose drift callables
    need the adjusted coordinates on the host), or a grid described by its axes and generated on the device (mik_set_grid:
    the meshgrid and the anisotropy adjustment of ok.py:863-885 never exist on the host)."""

    __slots__ = ("arrays", "axes", "center", "rot", "stretch", "mask", "extra", "shape", "npt", "raw")

    def __init__(self, shape, mask, extra, arrays=None, axes=None, center=None, rot=None, stretch=None, raw=False):
        self.shape, self.mask, self.extra = shape, mask, extra
        self.arrays, self.axes, self.center, self.rot, self.stretch = arrays, axes, center, rot, stretch
        self.raw = raw  # `arrays` are the coordinates as given: the device applies center / rot / stretch (mik_adjust_points)
        self.npt = int(np.prod(shape))

    def load(self, h, ndim, cell_range=None, with_extra=True):
        """H2D (arrays) or device-side generation (grid) of the points, optionally only cells [first, first + count).  The library
        stages host arrays synchronously (mik_set_points returns after its copy into page-locked memory): views of the caller's
        arrays (_Cols) are not referenced after this call returns, and must not be if the upload ever becomes asynchronous."""
        extra = self.extra if with_extra else None
        if cell_range is None:
            mask = self.mask
        else:
            sl = slice(cell_range[0], cell_range[0] + cell_range[1])
            mask = None if self.mask is None else self.mask[sl]
            extra = None if extra is None else np.ascontiguousarray(extra[:, sl])
        if self.axes is not None:
            h.set_grid(self.axes, self.center, self.rot, self.stretch, mask=mask, extra_rows=extra, cell_range=cell_range)
            return
        a = self.arrays if cell_range is None else self.arrays[sl]
        h.set_points(a[:, 0], a[:, 1], a[:, 2] if ndim == 3 else None, mask=mask, extra_rows=extra)
        if self.raw:
            h.adjust_points(self.center, self.rot, self.stretch)


class _KrigingBase:
    eps = 1.0e-10  # ok.py:177, uk.py:210
    UNBIAS = True  # uk.py:208
    # ok.py:178-185: the named models as functions (m, d) -> gamma; what `variogram_function` is for a named model.  The device
    # evaluates them itself (variogram_model selects the kernel); the functions are for callers that plot or inspect the fit.
    variogram_dict = {
        "linear": variogram_models.linear_variogram_model,
        "power": variogram_models.power_variogram_model,
        "gaussian": variogram_models.gaussian_variogram_model,
        "spherical": variogram_models.spherical_variogram_model,
        "exponential": variogram_models.exponential_variogram_model,
        "hole-effect": variogram_models.hole_effect_variogram_model,
    }
    _ndim = 2
    _universal = False
    _backends = ("vectorized", "loop", "hip")
    _label = "kriging"

    # ---------------------------------------------------------------- construction helpers
    def _init_common(self, variogram_model, variogram_parameters, variogram_function, exact_values, pseudo_inv,
                     pseudo_inv_type, verbose, enable_plotting):
        self.pseudo_inv = bool(pseudo_inv)
        self.pseudo_inv_type = str(pseudo_inv_type)
        if self.pseudo_inv_type not in ("pinv", "pinvh"):  # core.py:33 P_INV keys
            raise ValueError("pseudo inv type not valid: " + str(pseudo_inv_type))
        self.model = None
        if hasattr(variogram_model, "pykrige_kwargs"):
            # a GSTools CovModel (ok.py:223-239): it becomes a custom variogram; its anisotropy is taken by the caller
            self.model = variogram_model
            variogram_model, variogram_function, variogram_parameters = "custom", variogram_model.pykrige_vario, []
        self.variogram_model = variogram_model
        self.variogram_function = None
        if variogram_model == "custom":
            if variogram_function is None or not callable(variogram_function):
                raise ValueError("Must specify callable function for custom variogram model.")
            self.variogram_function = variogram_function  # evaluated on the host (it is Python), geometry stays on the device
        elif variogram_model not in core.MODELS:
            raise ValueError("Specified variogram model '%s' is not supported." % variogram_model)
        else:
            self.variogram_function = self.variogram_dict[variogram_model]  # ok.py:253
        if not isinstance(exact_values, bool):
            raise ValueError("exact_values has to be boolean True or False")
        self.exact_values = exact_values
        self.verbose = verbose
        self.enable_plotting = enable_plotting
        if self.enable_plotting and self.verbose:
            print("Plotting Enabled\n")
        self._user_parameters = variogram_parameters
        self._handle = None
        return variogram_parameters

    def _set_variogram_parameters(self, variogram_parameters, nlags, weight, updating=False):
        if self.verbose:  # (verbose=True narrates what upstream narrates, line for line: ok.py:310-375, 488-553 and the three siblings)
            print("Updating variogram mode..." if updating else "Initializing variogram model...")
        plist = core.make_variogram_parameter_list(self.variogram_model, variogram_parameters)
        fitted = plist is None
        if plist is None:
            from . import variogram_fit  # constructor-time only; never on the execute() path

            self.lags, self.semivariance, plist = variogram_fit.fit(
                self._coords_adj, self._values(), self.variogram_model, nlags, weight,
                getattr(self, "coordinates_type", "euclidean"), binned=self._device_variogram(nlags))
        else:  # the reference bins the experimental variogram even when parameters are given; here on first use
            self.__dict__.pop("lags", None)
            self.__dict__.pop("semivariance", None)
        self._nlags = nlags
        for name in self._LAZY_STATS:  # a new variogram invalidates the statistics
            if self.__dict__.get(name, 0) is not None:
                self.__dict__.pop(name, None)
        # a FITTED parameter set is the least-squares solution as it comes, an ndarray (core.py:575-627 returns res.x); a given one is a list (core.py:353-357)
        self.variogram_model_parameters = np.array([float(v) for v in plist]) if fitted else [float(v) for v in plist]
        if self.verbose:
            par = self.variogram_model_parameters
            if type(self) is OrdinaryKriging:
                print("Coordinates type: '%s'" % self.coordinates_type, "\n")
            if self.variogram_model == "linear":
                print("Using '%s' Variogram Model" % "linear")
                print("Slope:", par[0])
                print("Nugget:", par[1], "\n")
            elif self.variogram_model == "power":
                print("Using '%s' Variogram Model" % "power")
                print("Scale:", par[0])
                print("Exponent:", par[1])
                print("Nugget:", par[2], "\n")
            elif self.variogram_model == "custom":
                print("Using Custom Variogram Model")
            else:
                print("Using '%s' Variogram Model" % self.variogram_model)
                print("Partial Sill:", par[0])
                print("Full Sill:", par[0] + par[2])
                print("Range:", par[1])
                print("Nugget:", par[2], "\n")
        if getattr(self, "enable_plotting", False):  # ok.py:355-356, 534-535
            self.display_variogram_model()
        if self.verbose:
            print("Calculating statistics on variogram model fit...")
            # upstream computes them here (always in update_variogram_model and in the UK / 3-D constructors, on request in OrdinaryKriging's) and
            # prints them; without verbose they stay lazy (_LAZY_STATS)
            if updating or type(self) is not OrdinaryKriging or getattr(self, "_enable_statistics", False):
                self._compute_statistics()

    def _device_variogram(self, nlags):
        """Experimental semivariogram on the GPU (mik_experimental_variogram) when one is visible and the O(N^2) pair
        reduction is worth a launch; None -> the caller bins on the host with SciPy (same numbers to 1e-12).  This is
        constructor-time work, outside the execute() path, so a host route is legitimate here."""
        if self._values().size < int(_os.environ.get("MIK_DEVICE_VARIOGRAM_MIN_N", "512")) or nlags > 64:
            return None
        try:
            if _lib.load().mik_device_count() < 1:
                return None
        except ImportError:
            return None
        h = self._get_handle()
        self._factor_key = None
        ca = self._coords_adj
        h.set_problem(ndim=self._ndim, xs=ca[:, 0], ys=ca[:, 1], zs=ca[:, 2] if self._ndim == 3 else None,
                      values=self._values(), model_id=0, params=[1.0, 0.0],
                      geographic=getattr(self, "coordinates_type", "euclidean") == "geographic")
        return h.experimental_variogram(nlags)

    def _update_variogram_model(self, variogram_model, variogram_parameters, variogram_function, nlags, weight, anisotropy):
        """Body shared by the four update_variogram_model signatures (ok.py:379-545, uk.py:630-760, ok3d.py:368-520,
        uk3d.py:452-600) without the statistics pass, which execute() never reads.  As in the reference the anisotropy
        keywords DEFAULT to isotropic: omitting them resets the object to scaling 1 / angle 0 and re-adjusts the stations
        (UniversalKriging's point_log wells keep their construction-time adjustment -- the reference does not touch them)."""
        self.model = None
        if hasattr(variogram_model, "pykrige_kwargs"):  # GSTools CovModel: its own anisotropy (ok.py:430-444, ok3d.py:432-445)
            self.model = variogram_model
            if self._ndim == 2:
                if self.model.field_dim == 3:
                    raise ValueError("GSTools: model dim is not 1 or 2")
                if self.model.latlon and getattr(self, "coordinates_type", "euclidean") == "euclidean":
                    raise ValueError("GSTools: latlon models require geographic coordinates")
                anisotropy = dict(anisotropy_scaling=self.model.pykrige_anis, anisotropy_angle=self.model.pykrige_angle)
            else:
                if self.model.field_dim < 3:
                    raise ValueError("GSTools: model dim is not 3")
                anisotropy = dict(anisotropy_scaling_y=self.model.pykrige_anis_y, anisotropy_scaling_z=self.model.pykrige_anis_z,
                                  anisotropy_angle_x=self.model.pykrige_angle_x, anisotropy_angle_y=self.model.pykrige_angle_y,
                                  anisotropy_angle_z=self.model.pykrige_angle_z)
            variogram_model, variogram_function, variogram_parameters = "custom", variogram_model.pykrige_vario, []
        if variogram_model == "custom":
            if variogram_function is None or not callable(variogram_function):
                raise ValueError("Must specify callable function for custom variogram model.")
            self.variogram_function = variogram_function
        elif variogram_model not in core.MODELS:
            raise ValueError("Specified variogram model '%s' is not supported." % variogram_model)
        else:
            self.variogram_function = self.variogram_dict[variogram_model]
        self.variogram_model = variogram_model
        if getattr(self, "coordinates_type", "euclidean") == "geographic":
            if anisotropy.get("anisotropy_scaling", 1.0) != 1.0 and anisotropy["anisotropy_scaling"] != self.anisotropy_scaling:
                warnings.warn("Anisotropy is not compatible with geographic coordinates. Ignoring user set anisotropy.",
                              UserWarning)  # ok.py:478-487
        elif any(v != getattr(self, k) for k, v in anisotropy.items()):
            if self.verbose:
                print("Adjusting data for anisotropy...")
            for k, v in anisotropy.items():
                setattr(self, k, v)
            self._adjust_stations()
        self._set_variogram_parameters(variogram_parameters, nlags, weight, updating=True)

    def update_variogram_model(self, variogram_model, variogram_parameters=None, variogram_function=None, nlags=6,
                               weight=False, anisotropy_scaling=1.0, anisotropy_angle=0.0):
        """Changes the variogram model and the anisotropy (ok.py:379-387, uk.py:630-639: same signature, same defaults)."""
        self._update_variogram_model(variogram_model, variogram_parameters, variogram_function, nlags, weight,
                                     dict(anisotropy_scaling=anisotropy_scaling, anisotropy_angle=anisotropy_angle))

    # ---------------------------------------------------------------- device plumbing
    def _get_handle(self):
        if self._handle is None:
            self._handle = _lib.acquire_handle()  # a parked one if there is (mik_create + mik_destroy: 20 - 30 ms; _lib.release_handle)
        return self._handle

    _max_points = 0  # most points one call of this object handed to its handle

    def _handle_bytes(self):
        """Device memory the handle holds at most after this object's calls, for the parking rule: matrix, its probe copy and panels (3 Mp^2) + the
        right-hand sides of two launches of <= 131 072 points (2 npt Mp) + stations and results."""
        n = int(np.size(self._values())) + 16
        mp = -(-n // 128) * 128
        npt = min(int(self._max_points), 131072)
        return 8.0 * (3.0 * mp * mp + 2.0 * npt * mp + 8.0 * self._max_points + 8.0 * n)

    def __getstate__(self):
        """Pickling / copy.deepcopy (sklearn's clone, joblib workers, cached models): upstream's objects are plain Python attributes and travel; here the
        device side -- the library handle and the note of what is factored in it -- stays behind and the copy builds its own on its first execute()."""
        state = dict(self.__dict__)
        state["_handle"] = None
        state.pop("_factor_key", None)
        return state

    def __del__(self):
        h = self.__dict__.get("_handle")
        if h is not None:
            self._handle = None
            try:
                _lib.release_handle(h, self._handle_bytes())
            except Exception:  # noqa: BLE001  (interpreter shutdown: the handle's own __del__ closes it)
                pass

    def _values(self):
        return self.VALUES if self._ndim == 3 else self.Z

    def _station_extra_cols(self):
        return None

    def _wells(self):
        return None

    def _regional_linear(self):
        return False

    def _set_problem(self, h, with_drift=True, values=None, pseudo_inv=False, exact_values=None, eps=None):
        """H2D of the stations / drift description."""
        self._factor_key = None  # whatever the handle held is replaced
        ca = self._coords_adj
        kw = dict(
            ndim=self._ndim, xs=ca[:, 0], ys=ca[:, 1], zs=ca[:, 2] if self._ndim == 3 else None,
            values=self._values() if values is None else values, model_id=_lib.MODEL_IDS[self.variogram_model],
            params=self.variogram_model_parameters, eps=self.eps if eps is None else eps,
            exact_values=self.exact_values if exact_values is None else exact_values,
            regional_linear=self._regional_linear() if with_drift else False,
            wells=self._wells() if with_drift else None,
            extra_cols=self._station_extra_cols() if with_drift else None,
            geographic=getattr(self, "coordinates_type", "euclidean") == "geographic",
        )
        if self.variogram_model == "custom":
            fn, par = self.variogram_function, self.variogram_model_parameters
            h.set_custom_variogram(lambda d: fn(par, d))
            kw["params"] = [0.0, 0.0, 0.0]
        if self.pseudo_inv and (with_drift or pseudo_inv):
            # ok.py:660-661: a_inv = P_INV[self.pseudo_inv_type](a) -- on the device (mik_problem.pseudo_inv)
            kw["pseudo_inv"] = {"pinv": 1, "pinvh": 2}[self.pseudo_inv_type]
        h.set_problem(**kw)

    def _problem_key(self):
        """Digest of everything mik_set_problem + mik_factor depend on; None = not cacheable (a user callable)."""
        if self.variogram_model == "custom" or _os.environ.get("MIK_FACTOR_CACHE", "1") == "0":
            return None
        import hashlib

        d = hashlib.blake2b(digest_size=16)
        for a in (self._coords_adj, self._values(), self._wells(), self._station_extra_cols()):
            d.update(b"-" if a is None else np.ascontiguousarray(a, dtype=np.float64).tobytes())
        d.update(repr((self._ndim, self.variogram_model, [float(v) for v in self.variogram_model_parameters], float(self.eps),
                       bool(self.exact_values), bool(self._regional_linear()), bool(self.pseudo_inv), self.pseudo_inv_type,
                       getattr(self, "coordinates_type", "euclidean"))).encode())
        return d.digest()

    def _upload_and_factor(self):
        """K1 + K2 on the device (inverse, or pseudo-inverse when pseudo_inv=True).  The reference re-assembles and re-inverts
        the matrix on every execute() (ok.py:898, 663); here the factored matrix stays on the device and is reused while the
        problem (stations, values, variogram, drift set-up) is unchanged -- a second execute() on new points only predicts."""
        h = self._get_handle()
        key = self._problem_key()
        if key is not None:
            key = (key, h.option_epoch)  # a changed library option (factor path, device group, ...) invalidates the factor too
        if key is not None and key == getattr(self, "_factor_key", None):
            self.factor_reused = True
            return h
        self._set_problem(h)
        h.factor()
        self._factor_key = key
        self.factor_reused = False
        return h

    def _solve(self, P):
        h = self._upload_and_factor()
        self._max_points = max(self._max_points, int(np.prod(P.shape)))
        P.load(h, self._ndim)
        h.predict()
        self.last_timing = h.timing()
        return h.get_results()

    def _prepare(self, style, axes, mask, specified_drift_arrays=None, backend="vectorized"):
        """Everything execute() does on the host before the solve -> _Pts.  Grids ('grid' / 'masked') are handed to the
        device as axes + the anisotropy matrices unless a drift callable needs the adjusted coordinates on the host
        (functional drifts) or MIK_DEVICE_GRID=0 asks for the host meshgrid."""
        self._point_dtype = self._narrow_dtype(axes)
        if (style not in ("grid", "masked") or _os.environ.get("MIK_DEVICE_GRID", "1") == "0"
                or getattr(self, "functional_drift", False)):
            # coordinate arrays.  Round 3: they too are adjusted on the device (mik_adjust_points; the host only stacks them) unless
            # a functional drift needs the adjusted coordinates here, the coordinates are geographic (no adjustment at all,
            # ok.py:892-896) or MIK_DEVICE_POINTS=0 asks for the host adjustment
            raw = (not getattr(self, "functional_drift", False) and getattr(self, "coordinates_type", "euclidean") != "geographic"
                   and _os.environ.get("MIK_DEVICE_POINTS", "1") != "0")
            pts, shape, mask, extra = self._prepare_points(style, axes, mask, specified_drift_arrays, backend, adjust=not raw)
            if not raw:
                return _Pts(shape, mask, extra, arrays=pts)
            rot, stretch = core.anisotropy_matrices(self._ndim, self._scaling(), self._angle())
            return _Pts(shape, mask, extra, arrays=self._as_centred(pts), center=self._center(), rot=rot, stretch=stretch, raw=True)
        axes, shape, mask = self._grid_from(style, axes, mask)
        extra = self._grid_rows(style, axes, shape, mask, specified_drift_arrays, backend)
        if getattr(self, "coordinates_type", "euclidean") == "geographic":
            return _Pts(shape, mask, extra, axes=axes)  # no anisotropy correction in spherical coordinates (ok.py:892-896)
        rot, stretch = core.anisotropy_matrices(self._ndim, self._scaling(), self._angle())
        return _Pts(shape, mask, extra, axes=self._as_centred(axes), center=self._center(), rot=rot, stretch=stretch)

    _point_dtype = None

    def _narrow_dtype(self, axes):
        """The reference never casts the prediction coordinates (ok.py:849-850: np.array(xpoints, copy=True)) and centres them IN PLACE
        (core.py:146 `X -= center`, X = np.vstack of the coordinate arrays): when every array is float32 (or float16) the centred
        coordinates are rounded to that type -- 1e-7 relative, 1.7e-7 on z in the differential run of scripts/edge_forms_vs_reference.py --
        before the rotation continues in float64.  Returns that dtype, or None when the arrays promote to float64 (integer arrays: the
        reference's in-place subtract raises UFuncTypeError; here they are kriged as float64).  Geographic coordinates are not centred."""
        if getattr(self, "coordinates_type", "euclidean") == "geographic":
            return None
        try:
            rt = np.result_type(*[a.dtype if isinstance(a, np.ndarray) else np.asarray(a).dtype for a in axes])
        except Exception:  # noqa: BLE001  (ragged / object input: the float64 cast that follows raises what it raises)
            return None
        return rt if rt.kind == "f" and rt.itemsize < 8 else None

    def _as_centred(self, pts, nd="points"):
        """pts (float64: a list of axes, an (n, d) array or _Cols) as the reference's narrower in-place centring leaves them:
        x -> float64(narrow(x - c)) + c.  (The `+ c`, undone by the adjustment's own `- c`, costs an ulp of float64: nine orders below the effect.)
        nd: the narrow dtype (default: the one of this call's prediction points, _narrow_dtype)."""
        if isinstance(nd, str):
            nd = self._point_dtype
        if nd is None:
            return pts
        cen = self._center()
        if isinstance(pts, list):
            return [(a - c).astype(nd).astype(np.float64) + c for a, c in zip(pts, cen)]
        cols = [(np.asarray(pts[:, k]) - c).astype(nd).astype(np.float64) + c for k, c in enumerate(cen)]
        return _Cols(cols) if isinstance(pts, _Cols) else np.stack(cols, axis=1)

    def _grid_rows(self, style, axes, shape, mask, specified_drift_arrays, backend):
        return None  # host-evaluated drift rows of a grid: universal kriging only

    def _prepare_points(self, style, axes, mask, specified_drift_arrays=None, backend="vectorized", adjust=True):
        """Everything execute() does on the host before the solve: returns (pts_adj, shape, mask, extra_rows); adjust=False
        leaves the anisotropy adjustment of the coordinates to the device (mik_adjust_points)."""
        pts, shape, mask = self._points_from(style, axes, mask, columns=not adjust)
        if getattr(self, "coordinates_type", "euclidean") == "geographic" or not adjust:
            return pts, shape, mask, None  # no anisotropy correction in spherical coordinates (ok.py:892-896)
        return core.adjust_for_anisotropy(self._as_centred(pts), self._center(), self._scaling(), self._angle()), shape, mask, None

    # ---------------------------------------------------------------- variogram-fit statistics (core.py:759-851)
    _LAZY_STATS = ("delta", "sigma", "epsilon", "Q1", "Q2", "cR")

    def _compute_statistics(self):
        """_find_statistics on the device (mik_statistics): station i kriged from stations 0..i-1 with the ORDINARY
        system (core._krige ignores drift terms), then delta / sigma / epsilon and Q1, Q2, cR as the reference."""
        h = self._get_handle()
        y = self._values()
        if self.pseudo_inv:
            # core.py:749-750: lstsq per growing subset (duplicated stations); no recursion exists for singular leading
            # systems, so each station is kriged through the device pseudo-inverse (core._statistics_pseudo_inv).
            # N - 1 small factorisations: on ONE device (a device group would exchange every one of them), and with the
            # rule core._krige applies whatever the object's exact_values says -- b is zeroed at the coincident station with
            # the fixed tolerance 1e-10 (core.py:728-745).
            full = self._coords_adj
            if h.n_devices > 1:
                h = _lib.Handle(h.device)
                h.set_devices(1)

            def subset(i):
                self._coords_adj = full[:i]
                try:
                    self._set_problem(h, with_drift=False, values=y[:i], pseudo_inv=True, exact_values=True, eps=1e-10)
                finally:
                    self._coords_adj = full

            par = self.variogram_model_parameters
            gamma = ((lambda d: self.variogram_function(par, d)) if self.variogram_model == "custom"
                     else (lambda d: core.variogram_value(self.variogram_model, par, d)))
            try:
                k, ss = core._statistics_pseudo_inv(h, subset, full, y, gamma,
                                                    getattr(self, "coordinates_type", "euclidean") == "geographic")
            finally:
                if h is not self._get_handle():
                    h.close()
                self._factor_key = None
        else:
            self._set_problem(h, with_drift=False)
            k, ss = h.statistics(y.size)
        self.delta, self.sigma = core._delta_sigma(y, k, ss, self.eps)
        self.epsilon = self.delta / self.sigma
        self.Q1 = abs(np.sum(self.epsilon) / (self.epsilon.shape[0] - 1))
        self.Q2 = np.sum(self.epsilon**2) / (self.epsilon.shape[0] - 1)
        self.cR = self.Q2 * np.exp(np.sum(np.log(self.sigma**2)) / self.sigma.shape[0])
        if self.verbose:
            print("Q1 =", self.Q1)
            print("Q2 =", self.Q2)
            print("cR =", self.cR, "\n")

    def __getattr__(self, name):
        # UK / 3-D constructors of the reference run _find_statistics unconditionally (uk.py:380, ok3d.py:352,
        # uk3d.py:380); here the same attributes are computed on first use instead of at construction.
        if name in _KrigingBase._LAZY_STATS and "_handle" in self.__dict__:
            self._compute_statistics()
            return self.__dict__[name]
        if name in ("lags", "semivariance") and "_coords_adj" in self.__dict__:
            from . import variogram_fit

            nl = self.__dict__.get("_nlags", 6)
            binned = self._device_variogram(nl)
            self.lags, self.semivariance = binned if binned is not None else variogram_fit.experimental_variogram(
                self._coords_adj, self._values(), nl, getattr(self, "coordinates_type", "euclidean"))
            return self.__dict__[name]
        raise AttributeError(name)

    def get_epsilon_residuals(self):
        return self.epsilon

    def get_statistics(self):
        return self.Q1, self.Q2, self.cR

    def print_statistics(self):
        print("Q1 =", self.Q1)
        print("Q2 =", self.Q2)
        print("cR =", self.cR)

    def get_variogram_points(self):
        """(lags, variogram model evaluated at the lags) -- ok.py:569-587."""
        if self.variogram_model == "custom":
            return self.lags, self.variogram_function(self.variogram_model_parameters, self.lags)
        return self.lags, core.variogram_value(self.variogram_model, self.variogram_model_parameters, self.lags)

    def display_variogram_model(self):
        """Binned semivariances and the model curve (ok.py:555-567); needs matplotlib."""
        import matplotlib.pyplot as plt

        lags, model = self.get_variogram_points()
        ax = plt.figure().add_subplot(111)
        ax.plot(lags, self.semivariance, "r*")
        ax.plot(lags, model, "k-")
        plt.show()

    def plot_epsilon_residuals(self):
        """Scatter of the variogram-fit epsilon residuals (ok.py:601-609); needs matplotlib."""
        import matplotlib.pyplot as plt

        eps = self.epsilon
        ax = plt.figure().add_subplot(111)
        ax.scatter(range(eps.size), eps, c="k", marker="*")
        ax.axhline(y=0.0)
        plt.show()

    def _get_kriging_matrix(self, n=None, n_withdrifts=None):
        """The kriging matrix as the reference's method of the same name returns it (ok.py:626-648, uk.py:861-920,
        3-D twins), assembled by K1 on the device and copied back -- for inspection; execute() never brings it to the host.
        (n, and n_withdrifts of the universal classes, are upstream's positional arguments: the station count and the count with drift
        columns -- both follow from the object and are accepted for the call's sake.)"""
        h = self._get_handle()
        self._set_problem(h)
        h.assemble_only()
        return h.get_matrix(0)

    def switch_verbose(self):
        self.verbose = not self.verbose

    def switch_plotting(self):
        self.enable_plotting = not self.enable_plotting

    # ---------------------------------------------------------------- execute front / back matter
    _mw_backends = ()  # backends the reference accepts together with n_closest_points

    def _check_backend(self, backend, n_closest_points):
        if n_closest_points is not None and n_closest_points <= 1:
            raise ValueError("n_closest_points has to be at least two!")  # ok.py:837-840
        if backend not in self._backends:
            raise ValueError("Specified backend {} is not supported for {}.".format(backend, self._label))
        if n_closest_points is not None and backend not in self._mw_backends:
            raise ValueError("Specified backend {} for a moving window is not supported.".format(backend))  # ok.py:982-986

    def _solve_moving_window(self, P, n_closest_points, backend):
        """cKDTree.query + _exec_loop_moving_window / _c_exec_loop_moving_window on the device."""
        h = self._get_handle()
        self._set_problem(h)
        self._max_points = max(self._max_points, int(np.prod(P.shape)))
        P.load(h, self._ndim, with_extra=False)
        try:
            h.predict_moving_window(int(n_closest_points))
        except np.linalg.LinAlgError as e:
            if backend == "C":  # lib/cok.pyx:176-177 raises ValueError('Singular matrix'); scipy.linalg.solve LinAlgError
                raise ValueError("Singular matrix") from e
            raise
        self.last_timing = h.timing()
        return h.get_results()

    def _grid_from(self, style, axes, mask):
        """Axes cast to fp64, output shape and the flattened mask of style='grid' / 'masked' (ok.py:849-862, 896;
        ok3d.py:841-861, 893) -- the checks of the reference without its meshgrid."""
        # the reference does not cast (integer grids crash in its in-place subtract); cast to fp64 here
        axes = [np.atleast_1d(np.squeeze(np.array(a, copy=True, dtype=np.float64))) for a in axes]
        sizes = [a.size for a in axes]
        shape = tuple(reversed(sizes))  # (ny, nx) / (nz, ny, nx)
        if style == "masked":
            if mask is None:
                raise IOError("Must specify boolean masking array when style is 'masked'.")
            mask = np.asarray(mask)
            if self._ndim == 3 and mask.ndim != 3:
                raise ValueError("Mask is not three-dimensional.")
            if mask.ndim != self._ndim:
                raise ValueError("Mask dimensions do not match specified grid dimensions.")
            if mask.shape != shape:
                if mask.shape == tuple(sizes):
                    mask = mask.T if self._ndim == 2 else mask.swapaxes(0, 2)
                else:
                    raise ValueError("Mask dimensions do not match specified grid dimensions.")
            mask = mask.flatten().astype(bool)
        else:
            mask = None  # ok.py:896 / ok3d.py:893: `if style != "masked": mask = np.zeros(npt, dtype="bool")` -- a mask handed
            # to style="grid" is ignored, every cell is kriged
        return axes, shape, mask

    def _meshgrid(self, axes):
        """The reference's flattened meshgrid (ok.py:863-867, ok3d.py:866-870) as an (npt, d) array."""
        if self._ndim == 2:
            gx, gy = np.meshgrid(axes[0], axes[1])
            return np.stack((gx.ravel(), gy.ravel()), axis=1)
        gz, gy, gx = np.meshgrid(axes[2], axes[1], axes[0], indexing="ij")
        return np.stack((gx.ravel(), gy.ravel(), gz.ravel()), axis=1)

    def _points_from(self, style, axes, mask, columns=False):
        """Reference meshgrid order and mask handling (ok.py:849-878; ok3d.py:841-876).  columns=True (style='points' whose
        coordinates go to the device raw): the coordinate arrays stay the caller's (_Cols: no host copy of npt x d doubles;
        round 3 made one F-ordered copy, rounds 1-2 two copies)."""
        if style != "grid" and style != "masked" and style != "points":
            raise ValueError("style argument must be 'grid', 'points', or 'masked'")
        if style in ("grid", "masked"):
            axes, shape, mask = self._grid_from(style, axes, mask)
            pts = self._meshgrid(axes)
        else:
            # columns (the coordinates go to the device as they are): no host copy at all -- views of the caller's arrays (_Cols)
            axes = [np.atleast_1d(np.squeeze(np.asarray(a, dtype=np.float64) if columns else np.array(a, copy=True, dtype=np.float64)))
                    for a in axes]
            sizes = [a.size for a in axes]
            if len(set(sizes)) != 1:
                raise ValueError("xpoints and ypoints%s must have same dimensions when treated as listing "
                                 "discrete points." % (", zpoints" if self._ndim == 3 else ""))
            if any(a.ndim > 1 for a in axes):
                # the reference stacks xpts[:, np.newaxis] columns (ok.py:886-888) and hands them to cdist, which refuses anything but (n, d)
                raise ValueError("XB must be a 2-dimensional array.")
            shape = (sizes[0],)
            if columns:
                pts = _Cols([a.reshape(-1) for a in axes])
            else:
                pts = np.stack(axes, axis=1)
            mask = None
        return pts, shape, mask

    def _finish(self, z, ss, style, shape, mask, backend):
        if style == "masked":
            z = np.ma.array(z, mask=mask)
            ss = np.ma.array(ss, mask=mask)
        elif backend == "vectorized" and z.size:
            # reference quirk: _exec_vector returns MaskedArrays (all-False mask) in every style (an EMPTY point list comes back as plain arrays)
            z = np.ma.array(z, mask=np.zeros(z.shape, dtype=bool))
            ss = np.ma.array(ss, mask=np.zeros(ss.shape, dtype=bool))
        return z.reshape(shape), ss.reshape(shape)

    def _spec_rows(self, style, shape, npt, specified_drift_arrays):
        """uk.py:1217-1274 / uk3d.py:1030-1095: validate + flatten the per-point specified-drift arrays."""
        if specified_drift_arrays is None:
            specified_drift_arrays = []
        rows = []
        if self.specified_drift:
            if len(specified_drift_arrays) == 0:
                raise ValueError("Must provide drift values for kriging points when using 'specified' drift capability.")
            if type(specified_drift_arrays) is not list:
                raise TypeError("Arrays for specified drift terms must be encapsulated in a list.")
            for spec in specified_drift_arrays:
                spec = np.asarray(spec)
                if style in ("grid", "masked"):
                    if spec.ndim < self._ndim:
                        raise ValueError("Dimensions of drift values array do not match specified grid dimensions.")
                    if spec.shape != shape:
                        if spec.shape == tuple(reversed(shape)):
                            spec = spec.T if self._ndim == 2 else spec.swapaxes(0, 2)
                        else:
                            raise ValueError("Dimensions of drift values array do not match specified grid dimensions.")
                else:
                    if spec.ndim != 1:
                        raise ValueError("Dimensions of drift values array do not match specified grid dimensions.")
                    if spec.shape[0] != npt:
                        raise ValueError("Number of supplied drift values in array do not match specified number of kriging points.")
                rows.append(np.asarray(spec, dtype=np.float64).ravel())
            if len(rows) != len(self.specified_drift_data_arrays):
                raise ValueError("Inconsistent number of specified drift terms supplied.")
        elif len(specified_drift_arrays) != 0:
            warnings.warn("Provided specified drift values, but 'specified' drift was not initialized during "
                          "instantiation of %s class." % type(self).__name__, RuntimeWarning)
        return rows


# =====================================================================================================
class OrdinaryKriging(_KrigingBase):
    """2D ordinary kriging (ok.py:41).  ``execute`` runs on the GPU."""

    _ndim = 2
    _backends = ("vectorized", "loop", "C", "hip")
    _mw_backends = ("loop", "C", "hip")
    _label = "2D ordinary kriging"

    def __init__(self, x, y, z, variogram_model="linear", variogram_parameters=None, variogram_function=None, nlags=6,
                 weight=False, anisotropy_scaling=1.0, anisotropy_angle=0.0, verbose=False, enable_plotting=False,
                 enable_statistics=False, coordinates_type="euclidean", exact_values=True, pseudo_inv=False,
                 pseudo_inv_type="pinv"):
        variogram_parameters = self._init_common(variogram_model, variogram_parameters, variogram_function, exact_values,
                                                 pseudo_inv, pseudo_inv_type, verbose, enable_plotting)
        if self.model is not None:  # GSTools CovModel: 2-D only, its own anisotropy (ok.py:228-239)
            if self.model.field_dim == 3:
                raise ValueError("GSTools: model dim is not 1 or 2")
            if self.model.latlon and coordinates_type == "euclidean":
                raise ValueError("GSTools: latlon models require geographic coordinates")
            anisotropy_scaling, anisotropy_angle = self.model.pykrige_anis, self.model.pykrige_angle
        if coordinates_type not in ("euclidean", "geographic"):
            raise ValueError("Only 'euclidean' and 'geographic' are valid values for coordinates-keyword.")
        self.coordinates_type = coordinates_type
        self._enable_statistics = bool(enable_statistics)
        self.X_ORIG = np.atleast_1d(np.squeeze(np.array(x, copy=True, dtype=np.float64)))
        self.Y_ORIG = np.atleast_1d(np.squeeze(np.array(y, copy=True, dtype=np.float64)))
        self.Z = np.atleast_1d(np.squeeze(np.array(z, copy=True, dtype=np.float64)))
        if self.coordinates_type == "euclidean":
            self.XCENTER = (np.amax(self.X_ORIG) + np.amin(self.X_ORIG)) / 2.0
            self.YCENTER = (np.amax(self.Y_ORIG) + np.amin(self.Y_ORIG)) / 2.0
            self.anisotropy_scaling = anisotropy_scaling
            self.anisotropy_angle = anisotropy_angle
            if self.verbose:
                print("Adjusting data for anisotropy...")
            self._adjust_stations()
        else:  # ok.py:289-304: coordinates stay as they are; anisotropy is ambiguous on the sphere
            if anisotropy_scaling != 1.0:
                warnings.warn("Anisotropy is not compatible with geographic coordinates. Ignoring user set anisotropy.",
                              UserWarning)
            self.XCENTER = self.YCENTER = 0.0
            self.anisotropy_scaling, self.anisotropy_angle = 1.0, 0.0
            self.X_ADJUSTED, self.Y_ADJUSTED = self.X_ORIG, self.Y_ORIG
            self._coords_adj = np.vstack((self.X_ORIG, self.Y_ORIG)).T
        self._set_variogram_parameters(variogram_parameters, nlags, weight)
        if type(self) is OrdinaryKriging and not self._enable_statistics:  # ok.py:360-377: statistics only on request
            self.delta = self.sigma = self.epsilon = self.Q1 = self.Q2 = self.cR = None
        elif type(self) is OrdinaryKriging and self.__dict__.get("Q1") is None:  # (verbose: _set_variogram_parameters has computed and printed them)
            self._compute_statistics()

    def _center(self):
        return [self.XCENTER, self.YCENTER]

    def _scaling(self):
        return [self.anisotropy_scaling]

    def _angle(self):
        return [self.anisotropy_angle]

    def _adjust_stations(self):
        self._coords_adj = core.adjust_for_anisotropy(np.vstack((self.X_ORIG, self.Y_ORIG)).T, self._center(),
                                                      self._scaling(), self._angle())
        self.X_ADJUSTED, self.Y_ADJUSTED = self._coords_adj.T

    def execute(self, style, xpoints, ypoints, mask=None, backend="vectorized", n_closest_points=None):
        """Kriged values and variances at a grid / masked grid / list of points (ok.py:760-1020)."""
        if self.verbose:
            print("Executing Ordinary Kriging...\n")
        if style != "grid" and style != "masked" and style != "points":
            raise ValueError("style argument must be 'grid', 'points', or 'masked'")
        self._check_backend(backend, n_closest_points)
        P = self._prepare(style, (xpoints, ypoints), mask)
        if n_closest_points is not None:
            z, ss = self._solve_moving_window(P, n_closest_points, backend)
        else:
            z, ss = self._solve(P)
        return self._finish(z, ss, style, P.shape, P.mask, backend)


# =====================================================================================================
class UniversalKriging(OrdinaryKriging):
    """2D universal kriging (uk.py:39): regional_linear and point_log drifts are evaluated on the
    device; external_Z, specified and functional drifts are evaluated here (they are O(npt) lookups or
    user callables) and passed as extra matrix columns / RHS rows."""

    _universal = True
    _backends = ("vectorized", "loop", "hip")
    _mw_backends = ()
    _label = "2D universal kriging"

    def __init__(self, x, y, z, variogram_model="linear", variogram_parameters=None, variogram_function=None, nlags=6,
                 weight=False, anisotropy_scaling=1.0, anisotropy_angle=0.0, drift_terms=None, point_drift=None,
                 external_drift=None, external_drift_x=None, external_drift_y=None, specified_drift=None,
                 functional_drift=None, verbose=False, enable_plotting=False, exact_values=True, pseudo_inv=False,
                 pseudo_inv_type="pinv"):
        OrdinaryKriging.__init__(self, x, y, z, variogram_model=variogram_model,
                                 variogram_parameters=variogram_parameters, variogram_function=variogram_function,
                                 nlags=nlags, weight=weight, anisotropy_scaling=anisotropy_scaling,
                                 anisotropy_angle=anisotropy_angle, verbose=verbose, enable_plotting=enable_plotting,
                                 exact_values=exact_values, pseudo_inv=pseudo_inv, pseudo_inv_type=pseudo_inv_type)
        if drift_terms is None:
            drift_terms = []
        if specified_drift is None:  # uk.py:254-257, uk3d.py:249-252: an omitted list is an EMPTY list (-> ValueError, not TypeError)
            specified_drift = []
        if functional_drift is None:
            functional_drift = []
        if self.verbose:  # uk.py:396-466
            print("Initializing drift terms...")
        self.regional_linear_drift = "regional_linear" in drift_terms
        if self.regional_linear_drift and self.verbose:
            print("Implementing regional linear drift.")
        self.external_Z_drift = "external_Z" in drift_terms
        if self.external_Z_drift:
            if external_drift is None:
                raise ValueError("Must specify external Z drift terms.")
            if external_drift_x is None or external_drift_y is None:
                raise ValueError("Must specify coordinates of external Z drift terms.")
            external_drift = np.asarray(external_drift)
            ex, ey = np.asarray(external_drift_x), np.asarray(external_drift_y)
            if external_drift.shape[0] != ey.shape[0] or external_drift.shape[1] != ex.shape[0]:
                if external_drift.shape[0] == ex.shape[0] and external_drift.shape[1] == ey.shape[0]:
                    self.external_Z_array = np.array(external_drift.T)
                else:
                    raise ValueError("External drift dimensions do not match provided x- and y-coordinate dimensions.")
            else:
                self.external_Z_array = np.array(external_drift)
            self.external_Z_array_x = np.array(ex).flatten()
            self.external_Z_array_y = np.array(ey).flatten()
            self.z_scalars = self._calculate_data_point_zscalars(self.X_ORIG, self.Y_ORIG)
            if self.verbose:
                print("Implementing external Z drift.")
        self.point_log_drift = "point_log" in drift_terms
        if self.point_log_drift:
            if point_drift is None:
                raise ValueError("Must specify location(s) and strength(s) of point drift terms.")
            raw = np.asarray(point_drift)
            # (the wells go through the same in-place centring as the prediction points, uk.py:450-459: a float32 point_drift array is rounded there)
            self._point_log_dtype = raw.dtype if raw.dtype.kind == "f" and raw.dtype.itemsize < 8 else None
            point_log = np.atleast_2d(np.squeeze(np.array(point_drift, copy=True, dtype=np.float64)))
            self._point_log_user = point_log
            self._adjust_wells()
            if self.verbose:
                print("Implementing external point-logarithmic drift; number of points =", self.point_log_array.shape[0], "\n")
        self.specified_drift = "specified" in drift_terms
        if self.specified_drift:
            if type(specified_drift) is not list:
                raise TypeError("Arrays for specified drift terms must be encapsulated in a list.")
            if len(specified_drift) == 0:
                raise ValueError("Must provide at least one drift-value array when using the 'specified' drift capability.")
            self.specified_drift_data_arrays = []
            for term in specified_drift:
                specified = np.squeeze(np.array(term, copy=True))
                if specified.size != self.X_ORIG.size:
                    raise ValueError("Must specify the drift values for each data point when using the "
                                     "'specified' drift capability.")
                self.specified_drift_data_arrays.append(specified)
        self.functional_drift = "functional" in drift_terms
        if self.functional_drift:
            if type(functional_drift) is not list:
                raise TypeError("Callables for functional drift terms must be encapsulated in a list.")
            if len(functional_drift) == 0:
                raise ValueError("Must provide at least one callable object when using the 'functional' drift capability.")
            self.functional_drift_terms = functional_drift

    def _adjust_wells(self):
        pl = self._point_log_user
        self.point_log_array = np.zeros(pl.shape)
        self.point_log_array[:, 2] = pl[:, 2]
        self.point_log_array[:, :2] = core.adjust_for_anisotropy(self._as_centred(np.vstack((pl[:, 0], pl[:, 1])).T, getattr(self, "_point_log_dtype", None)),
                                                                 self._center(), self._scaling(), self._angle())

    def _calculate_data_point_zscalars(self, x, y, type_="array"):
        return core.bilinear_zscalars(self.external_Z_array, self.external_Z_array_x, self.external_Z_array_y, x, y)

    def _regional_linear(self):
        return self.regional_linear_drift

    def _wells(self):
        return self.point_log_array if self.point_log_drift else None

    def _station_extra_cols(self):
        cols = []  # reference order after regional_linear and point_log: external_Z, specified, functional
        if self.external_Z_drift:
            cols.append(self.z_scalars)
        if self.specified_drift:
            cols.extend(self.specified_drift_data_arrays)
        if self.functional_drift:
            cols.extend(f(self.X_ADJUSTED, self.Y_ADJUSTED) for f in self.functional_drift_terms)
        return np.array(cols, dtype=np.float64) if cols else None

    def execute(self, style, xpoints, ypoints, mask=None, backend="vectorized", specified_drift_arrays=None):
        """uk.py:1090-1328."""
        if self.verbose:
            print("Executing Universal Kriging...\n")
        if style != "grid" and style != "masked" and style != "points":
            raise ValueError("style argument must be 'grid', 'points', or 'masked'")
        self._check_backend(backend, None)
        P = self._prepare(style, (xpoints, ypoints), mask, specified_drift_arrays, backend)
        z, ss = self._solve(P)
        return self._finish(z, ss, style, P.shape, P.mask, backend)

    def _grid_rows(self, style, axes, shape, mask, specified_drift_arrays, backend):
        rows = []
        if self.external_Z_drift:  # looked up at the ORIGINAL coordinates (uk.py:967-971): the host meshgrid, for this term only
            pts = self._meshgrid(axes)
            if mask is not None and backend != "vectorized":  # see _prepare_points
                zs = np.zeros(pts.shape[0])
                keep = ~mask
                zs[keep] = self._calculate_data_point_zscalars(pts[keep, 0], pts[keep, 1])
                rows.append(zs)
            else:
                rows.append(self._calculate_data_point_zscalars(pts[:, 0], pts[:, 1]))
        rows.extend(self._spec_rows(style, shape, int(np.prod(shape)), specified_drift_arrays))
        return np.array(rows, dtype=np.float64) if rows else None

    def _prepare_points(self, style, axes, mask, specified_drift_arrays=None, backend="vectorized", adjust=True):
        pts, shape, mask = self._points_from(style, axes, mask, columns=not adjust)
        rows = []
        if self.external_Z_drift:  # on ORIGINAL coordinates (uk.py:967-971)
            if mask is not None and backend != "vectorized":
                # the reference's loop looks the drift up point by point and skips masked points (uk.py:1034, 1061-1066), so a
                # masked point outside the drift grid does not raise there; 'vectorized' looks every point up (uk.py:967-971)
                zs = np.zeros(pts.shape[0])
                keep = ~mask
                zs[keep] = self._calculate_data_point_zscalars(pts[keep, 0], pts[keep, 1])
                rows.append(zs)
            else:
                rows.append(self._calculate_data_point_zscalars(pts[:, 0], pts[:, 1]))
        rows.extend(self._spec_rows(style, shape, pts.shape[0], specified_drift_arrays))
        if not adjust:  # (never with functional drifts: _prepare) the device adjusts the coordinates
            return pts, shape, mask, (np.array(rows, dtype=np.float64) if rows else None)
        pts_adj = core.adjust_for_anisotropy(self._as_centred(pts), self._center(), self._scaling(), self._angle())
        if self.functional_drift:
            rows.extend(np.asarray(f(pts_adj[:, 0], pts_adj[:, 1]), dtype=np.float64) for f in self.functional_drift_terms)
        return pts_adj, shape, mask, (np.array(rows, dtype=np.float64) if rows else None)


# =====================================================================================================
class OrdinaryKriging3D(_KrigingBase):
    """3D ordinary kriging (ok3d.py:40)."""

    _ndim = 3
    _mw_backends = ("loop", "hip")
    _label = "3D ordinary kriging"

    def __init__(self, x, y, z, val, variogram_model="linear", variogram_parameters=None, variogram_function=None,
                 nlags=6, weight=False, anisotropy_scaling_y=1.0, anisotropy_scaling_z=1.0, anisotropy_angle_x=0.0,
                 anisotropy_angle_y=0.0, anisotropy_angle_z=0.0, verbose=False, enable_plotting=False,
                 exact_values=True, pseudo_inv=False, pseudo_inv_type="pinv"):
        variogram_parameters = self._init_common(variogram_model, variogram_parameters, variogram_function, exact_values,
                                                 pseudo_inv, pseudo_inv_type, verbose, enable_plotting)
        if self.model is not None:  # GSTools CovModel: 3-D only here, its own anisotropy (ok3d.py:232-245)
            if self.model.field_dim < 3:
                raise ValueError("GSTools: model dim is not 3")
            anisotropy_scaling_y, anisotropy_scaling_z = self.model.pykrige_anis_y, self.model.pykrige_anis_z
            anisotropy_angle_x, anisotropy_angle_y, anisotropy_angle_z = (self.model.pykrige_angle_x, self.model.pykrige_angle_y,
                                                                          self.model.pykrige_angle_z)
        self.X_ORIG = np.atleast_1d(np.squeeze(np.array(x, copy=True, dtype=np.float64)))
        self.Y_ORIG = np.atleast_1d(np.squeeze(np.array(y, copy=True, dtype=np.float64)))
        self.Z_ORIG = np.atleast_1d(np.squeeze(np.array(z, copy=True, dtype=np.float64)))
        self.VALUES = np.atleast_1d(np.squeeze(np.array(val, copy=True, dtype=np.float64)))
        self.XCENTER = (np.amax(self.X_ORIG) + np.amin(self.X_ORIG)) / 2.0
        self.YCENTER = (np.amax(self.Y_ORIG) + np.amin(self.Y_ORIG)) / 2.0
        self.ZCENTER = (np.amax(self.Z_ORIG) + np.amin(self.Z_ORIG)) / 2.0
        self.anisotropy_scaling_y = anisotropy_scaling_y
        self.anisotropy_scaling_z = anisotropy_scaling_z
        self.anisotropy_angle_x = anisotropy_angle_x
        self.anisotropy_angle_y = anisotropy_angle_y
        self.anisotropy_angle_z = anisotropy_angle_z
        if self.verbose:
            print("Adjusting data for anisotropy...")
        self._adjust_stations()
        self._set_variogram_parameters(variogram_parameters, nlags, weight)

    def _center(self):
        return [self.XCENTER, self.YCENTER, self.ZCENTER]

    def _scaling(self):
        return [self.anisotropy_scaling_y, self.anisotropy_scaling_z]

    def _angle(self):
        return [self.anisotropy_angle_x, self.anisotropy_angle_y, self.anisotropy_angle_z]

    def _adjust_stations(self):
        self._coords_adj = core.adjust_for_anisotropy(np.vstack((self.X_ORIG, self.Y_ORIG, self.Z_ORIG)).T,
                                                      self._center(), self._scaling(), self._angle())
        self.X_ADJUSTED, self.Y_ADJUSTED, self.Z_ADJUSTED = self._coords_adj.T

    def update_variogram_model(self, variogram_model, variogram_parameters=None, variogram_function=None, nlags=6,
                               weight=False, anisotropy_scaling_y=1.0, anisotropy_scaling_z=1.0, anisotropy_angle_x=0.0,
                               anisotropy_angle_y=0.0, anisotropy_angle_z=0.0):
        """ok3d.py:368-380 / uk3d.py:452-464: same signature, same (isotropic) defaults."""
        self._update_variogram_model(variogram_model, variogram_parameters, variogram_function, nlags, weight,
                                     dict(anisotropy_scaling_y=anisotropy_scaling_y, anisotropy_scaling_z=anisotropy_scaling_z,
                                          anisotropy_angle_x=anisotropy_angle_x, anisotropy_angle_y=anisotropy_angle_y,
                                          anisotropy_angle_z=anisotropy_angle_z))

    def execute(self, style, xpoints, ypoints, zpoints, mask=None, backend="vectorized", n_closest_points=None):
        """ok3d.py:735-932.  Output shape (nz, ny, nx) for grids."""
        if self.verbose:
            print("Executing Ordinary Kriging...\n")
        if style != "grid" and style != "masked" and style != "points":
            raise ValueError("style argument must be 'grid', 'points', or 'masked'")
        self._check_backend(backend, n_closest_points)
        P = self._prepare(style, (xpoints, ypoints, zpoints), mask)
        if n_closest_points is not None:
            z, ss = self._solve_moving_window(P, n_closest_points, backend)
        else:
            z, ss = self._solve(P)
        return self._finish(z, ss, style, P.shape, P.mask, backend)


# =====================================================================================================
class UniversalKriging3D(OrdinaryKriging3D):
    """3D universal kriging (uk3d.py:39): regional_linear on the device, specified / functional drifts
    evaluated on the host."""

    _universal = True
    _mw_backends = ()
    _label = "3D universal kriging"

    def __init__(self, x, y, z, val, variogram_model="linear", variogram_parameters=None, variogram_function=None,
                 nlags=6, weight=False, anisotropy_scaling_y=1.0, anisotropy_scaling_z=1.0, anisotropy_angle_x=0.0,
                 anisotropy_angle_y=0.0, anisotropy_angle_z=0.0, drift_terms=None, specified_drift=None,
                 functional_drift=None, verbose=False, enable_plotting=False, exact_values=True, pseudo_inv=False,
                 pseudo_inv_type="pinv"):
        OrdinaryKriging3D.__init__(self, x, y, z, val, variogram_model=variogram_model,
                                   variogram_parameters=variogram_parameters, variogram_function=variogram_function,
                                   nlags=nlags, weight=weight, anisotropy_scaling_y=anisotropy_scaling_y,
                                   anisotropy_scaling_z=anisotropy_scaling_z, anisotropy_angle_x=anisotropy_angle_x,
                                   anisotropy_angle_y=anisotropy_angle_y, anisotropy_angle_z=anisotropy_angle_z,
                                   verbose=verbose, enable_plotting=enable_plotting, exact_values=exact_values,
                                   pseudo_inv=pseudo_inv, pseudo_inv_type=pseudo_inv_type)
        if drift_terms is None:
            drift_terms = []
        if specified_drift is None:  # uk.py:254-257, uk3d.py:249-252: an omitted list is an EMPTY list (-> ValueError, not TypeError)
            specified_drift = []
        if functional_drift is None:
            functional_drift = []
        if self.verbose:  # uk3d.py:396-406
            print("Initializing drift terms...")
            if "regional_linear" in drift_terms:
                print("Implementing regional linear drift.")
        self.regional_linear_drift = "regional_linear" in drift_terms
        self.specified_drift = "specified" in drift_terms
        if self.specified_drift:
            if type(specified_drift) is not list:
                raise TypeError("Arrays for specified drift terms must be encapsulated in a list.")
            if len(specified_drift) == 0:
                raise ValueError("Must provide at least one drift-value array when using the 'specified' drift capability.")
            self.specified_drift_data_arrays = []
            for term in specified_drift:
                specified = np.squeeze(np.array(term, copy=True))
                if specified.size != self.X_ORIG.size:
                    raise ValueError("Must specify the drift values for each data point when using the "
                                     "'specified' drift capability.")
                self.specified_drift_data_arrays.append(specified)
        self.functional_drift = "functional" in drift_terms
        if self.functional_drift:
            if type(functional_drift) is not list:
                raise TypeError("Callables for functional drift terms must be encapsulated in a list.")
            if len(functional_drift) == 0:
                raise ValueError("Must provide at least one callable object when using the 'functional' drift capability.")
            self.functional_drift_terms = functional_drift

    def _regional_linear(self):
        return self.regional_linear_drift

    def _station_extra_cols(self):
        cols = []
        if self.specified_drift:
            cols.extend(self.specified_drift_data_arrays)
        if self.functional_drift:
            cols.extend(f(self.X_ADJUSTED, self.Y_ADJUSTED, self.Z_ADJUSTED) for f in self.functional_drift_terms)
        return np.array(cols, dtype=np.float64) if cols else None

    def execute(self, style, xpoints, ypoints, zpoints, mask=None, backend="vectorized", specified_drift_arrays=None):
        """uk3d.py:877-1146."""
        if self.verbose:
            print("Executing Universal Kriging...\n")
        if style != "grid" and style != "masked" and style != "points":
            raise ValueError("style argument must be 'grid', 'points', or 'masked'")
        self._check_backend(backend, None)
        P = self._prepare(style, (xpoints, ypoints, zpoints), mask, specified_drift_arrays)
        z, ss = self._solve(P)
        return self._finish(z, ss, style, P.shape, P.mask, backend)

    def _grid_rows(self, style, axes, shape, mask, specified_drift_arrays, backend):
        rows = self._spec_rows(style, shape, int(np.prod(shape)), specified_drift_arrays)
        return np.array(rows, dtype=np.float64) if rows else None

    def _prepare_points(self, style, axes, mask, specified_drift_arrays=None, backend="vectorized", adjust=True):
        pts, shape, mask = self._points_from(style, axes, mask, columns=not adjust)
        rows = self._spec_rows(style, shape, pts.shape[0], specified_drift_arrays)
        if not adjust:  # (never with functional drifts: _prepare) the device adjusts the coordinates
            return pts, shape, mask, (np.array(rows, dtype=np.float64) if rows else None)
        pts_adj = core.adjust_for_anisotropy(self._as_centred(pts), self._center(), self._scaling(), self._angle())
        if self.functional_drift:
            rows.extend(np.asarray(f(pts_adj[:, 0], pts_adj[:, 1], pts_adj[:, 2]), dtype=np.float64)
                        for f in self.functional_drift_terms)
        return pts_adj, shape, mask, (np.array(rows, dtype=np.float64) if rows else None)
