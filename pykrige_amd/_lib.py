"""ctypes binding of libmikrige.so (include/mikrige.h).  NumPy + ctypes only -- no torch.

There is deliberately no CPU fallback here: if the HIP library is missing or no GPU is visible the
calls raise.  (oracle/ is test infrastructure and is never imported from this package.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MIK_LIB_PATH: a developer switch for A/B runs of two builds on one GPU box; the product loads the in-tree library)
LIB_PATH = os.environ.get("MIK_LIB_PATH") or os.path.join(_HERE, "libmikrige.so")

MIK_OK, MIK_EINVAL, MIK_ESINGULAR, MIK_EHIP, MIK_ERCCL, MIK_ESTATE = 0, -1, -2, -3, -4, -5
MODEL_IDS = {"linear": 0, "power": 1, "gaussian": 2, "spherical": 3, "exponential": 4, "hole-effect": 5, "custom": 6}
VARIOGRAM_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int64, C.c_int64)

_dp = C.POINTER(C.c_double)


class MikProblem(C.Structure):
    _fields_ = [
        ("ndim", C.c_int32), ("model_id", C.c_int32), ("n", C.c_int64),
        ("xs", _dp), ("ys", _dp), ("zs", _dp), ("values", _dp),
        ("params", C.c_double * 3), ("eps", C.c_double),
        ("exact_values", C.c_int32), ("regional_linear", C.c_int32), ("n_wells", C.c_int32), ("n_extra", C.c_int32),
        ("wells", _dp), ("extra_cols", _dp), ("a_inv", _dp),
        ("geographic", C.c_int32), ("pseudo_inv", C.c_int32),
    ]


class MikPoints(C.Structure):
    _fields_ = [
        ("npt", C.c_int64), ("px", _dp), ("py", _dp), ("pz", _dp),
        ("mask", C.POINTER(C.c_int8)), ("extra_rows", _dp),
    ]


class MikGrid(C.Structure):
    _fields_ = [
        ("ndim", C.c_int32), ("adjust", C.c_int32), ("nx", C.c_int64), ("ny", C.c_int64), ("nz", C.c_int64),
        ("gx", _dp), ("gy", _dp), ("gz", _dp),
        ("center", C.c_double * 3), ("rot", C.c_double * 9), ("stretch", C.c_double * 3),
        ("cell_first", C.c_int64), ("cell_count", C.c_int64),
        ("mask", C.POINTER(C.c_int8)), ("extra_rows", _dp),
    ]


class MikTiming(C.Structure):
    _fields_ = [
        ("assemble_ms", C.c_double), ("invert_ms", C.c_double), ("rhs_ms", C.c_double),
        ("contract_ms", C.c_double), ("predict_ms", C.c_double), ("contract_launches", C.c_int64),
        ("contract_flops_executed", C.c_double), ("factor_path", C.c_int32), ("symmetric", C.c_int32),
        ("engine", C.c_int32), ("reserved", C.c_int32),
        ("exchange_ms", C.c_double), ("exchange_path", C.c_int32), ("n_devices", C.c_int32),
        ("exchange_wait_ms", C.c_double), ("exchange_fallbacks", C.c_int32), ("rccl_ranks", C.c_int32),
        ("mw_kernel", C.c_int32), ("half_sweep", C.c_int32), ("factor_attempts", C.c_int32), ("null_dim", C.c_int32), ("rhs_overlapped", C.c_int32),
        ("verify_ms", C.c_double), ("verify_res_z", C.c_double), ("verify_res_inv", C.c_double),
        ("sparse", C.c_int32), ("stations_sorted", C.c_int32),
        ("sparse_tiles", C.c_double), ("sparse_tiles_dense", C.c_double),
        ("sparse_ktiles", C.c_double), ("sparse_ktiles_dense", C.c_double), ("sparse_lists_ms", C.c_double),
        ("sparse_diag_products", C.c_double), ("sparse_rows", C.c_int32), ("points_sorted", C.c_int32),
        ("sort_points_ms", C.c_double), ("sparse_ktile", C.c_int32), ("reserved2", C.c_int32), ("exchange_bytes", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None
ABI_VERSION = 7  # include/mikrige.h MIK_ABI_VERSION

# every entry point include/mikrige.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "mik_device_count": (C.c_int, []),
    "mik_abi_version": (C.c_int, []),
    "mik_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "mik_destroy": (None, [C.c_void_p]),
    "mik_set_devices": (C.c_int, [C.c_int]),
    "mik_handle_set_devices": (C.c_int, [C.c_void_p, C.c_int]),
    "mik_handle_devices": (C.c_int, [C.c_void_p]),
    "mik_slab_of": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mik_get_device_timing": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(MikTiming)]),
    "mik_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "mik_set_problem": (C.c_int, [C.c_void_p, C.POINTER(MikProblem)]),
    "mik_station_order": (C.c_int, [C.POINTER(MikProblem), C.POINTER(C.c_int32)]),
    "mik_factor": (C.c_int, [C.c_void_p]),
    "mik_set_points": (C.c_int, [C.c_void_p, C.POINTER(MikPoints)]),
    "mik_set_grid": (C.c_int, [C.c_void_p, C.POINTER(MikGrid)]),
    "mik_adjust_points": (C.c_int, [C.c_void_p, _dp, _dp, _dp]),
    "mik_predict": (C.c_int, [C.c_void_p]),
    "mik_get_results": (C.c_int, [C.c_void_p, _dp, _dp]),
    "mik_take_results": (C.c_int, [C.c_void_p, C.POINTER(_dp), C.POINTER(_dp)]),
    "mik_release_results": (None, [_dp]),
    "mik_synchronize": (C.c_int, [C.c_void_p]),
    "mik_set_custom_variogram": (C.c_int, [C.c_void_p, VARIOGRAM_FN, C.c_void_p]),
    "mik_predict_moving_window": (C.c_int, [C.c_void_p, C.c_int]),
    "mik_statistics": (C.c_int, [C.c_void_p, _dp, _dp]),
    "mik_experimental_variogram": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, C.POINTER(C.c_int32)]),
    "mik_krige_execute": (C.c_int, [C.c_int, C.POINTER(MikProblem), C.POINTER(MikPoints), _dp, _dp]),
    "mik_assemble_only": (C.c_int, [C.c_void_p]),
    "mik_get_matrix": (C.c_int, [C.c_void_p, C.c_int, _dp]),
    "mik_matrix_order": (C.c_int64, [C.c_void_p]),
    "mik_points_resident": (C.c_int64, [C.c_void_p]),
    "mik_get_points": (C.c_int, [C.c_void_p, _dp, _dp, _dp]),
    "mik_get_timing": (C.c_int, [C.c_void_p, C.POINTER(MikTiming)]),
    "mik_selftest_mfma": (C.c_int, [C.c_int]),
    "mik_selftest_exp": (C.c_int, [C.c_int, _dp, _dp, C.c_int]),
    "mik_comm_unique_id": (C.c_int, [C.c_char_p]),
    "mik_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p]),
    "mik_bcast_factor": (C.c_int, [C.c_void_p, C.c_int]),
    "mik_factor_checksum": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "mik_exchange_note": (C.c_char_p, [C.c_void_p]),
    "mik_selftest_exchange": (C.c_int, [C.c_int, C.c_double, C.c_double, C.c_char_p, C.c_int]),
    "mik_last_error": (C.c_char_p, []),
}


def load():
    """dlopen the library and declare the signatures.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "pykrige_amd: %s is missing -- build it with `python -m pykrige_amd.build` "
            "(needs hipcc; there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    # the version first: a library from before a symbol was added must end in the "rebuild it" message, not in a bare AttributeError
    have = getattr(lib, "mik_abi_version", None)
    version = None
    if have is not None:
        have.restype, have.argtypes = SIGNATURES["mik_abi_version"]
        version = have()
    stale = "pykrige_amd: %s %%s, this package was written against ABI version %d -- rebuild it (`python -m pykrige_amd.build --force`)" % (
        LIB_PATH, ABI_VERSION)
    if version != ABI_VERSION:
        raise ImportError(stale % ("has no mik_abi_version" if version is None else "has ABI version %d" % version))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise ImportError(stale % ("does not export %s" % name)) from None
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return (load().mik_last_error() or b"").decode("utf-8", "replace")


def check(rc):
    if rc == MIK_OK:
        return
    msg = last_error()
    if rc == MIK_EINVAL:
        raise ValueError(msg)
    if rc == MIK_ESINGULAR:
        raise np.linalg.LinAlgError(msg or "singular matrix")
    raise RuntimeError("libmikrige: %s (code %d)" % (msg, rc))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data_as(_dp) if a is not None else None


class _PinnedOwner:
    """Keeps a page-locked result buffer of the library alive for the NumPy arrays that view it; gives it back when they are gone."""

    def __init__(self, lib, ptr):
        self._lib, self._ptr = lib, ptr

    def __del__(self):
        try:
            self._lib.mik_release_results(self._ptr)
        except Exception:  # interpreter shutdown
            pass


class Handle:
    """A device context of the library: one HIP device, one stream, cached device buffers."""

    def __init__(self, device=None):
        lib = load()
        if device is None:
            device = int(os.environ.get("MIK_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            if lib.mik_device_count() == 1:
                device = 0
        self._lib = lib
        self._h = C.c_void_p()
        check(lib.mik_create(int(device), C.byref(self._h)))
        self.device = int(device)
        self._keep = []
        self.option_epoch = 0  # bumped by every option / device-group change (callers that cache a factor watch it)
        self.env_sig = _env_sig()  # the MIK_* environment the library read its defaults from when it created the handle
        self.pid = os.getpid()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            if getattr(self, "pid", os.getpid()) == os.getpid():  # (a forked child only forgets the parent's handle: its HIP objects are not the child's to destroy)
                self._lib.mik_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        self.option_epoch += 1
        check(self._lib.mik_set_option(self._h, key.encode(), float(value)))

    # --- single-process multi-GPU: the handle spans n devices (mik_handle_set_devices) -----------------
    def set_devices(self, n, alias=False):
        """Let this handle span `n` GPUs of the node (0 = all visible).  alias=True allows more members than GPUs (several
        logical devices on one physical GPU -- only useful to exercise the multi-device path on a 1-GPU box)."""
        if alias:
            self.set_option("alias_devices", 1)
        self.option_epoch += 1
        check(self._lib.mik_handle_set_devices(self._h, int(n)))

    @property
    def n_devices(self):
        return int(self._lib.mik_handle_devices(self._h))

    def device_timing(self, member):
        t = MikTiming()
        check(self._lib.mik_get_device_timing(self._h, int(member), C.byref(t)))
        return t.as_dict()

    def set_problem(self, ndim, xs, ys, zs, values, model_id, params, eps=1e-10, exact_values=True,
                    regional_linear=False, wells=None, extra_cols=None, a_inv=None, geographic=False, pseudo_inv=0):
        p = MikProblem()
        xs, ys, values = _f64(xs), _f64(ys), _f64(values)
        zs = _f64(zs) if zs is not None else None
        wells = _f64(wells).reshape(-1, 3) if wells is not None and np.size(wells) else None
        extra_cols = _f64(extra_cols).reshape(-1, xs.size) if extra_cols is not None and np.size(extra_cols) else None
        a_inv = _f64(a_inv) if a_inv is not None else None
        if not (xs.size == ys.size == values.size) or (zs is not None and zs.size != xs.size):
            raise ValueError("station arrays must have the same length")
        p.ndim, p.model_id, p.n = int(ndim), int(model_id), xs.size
        p.xs, p.ys, p.zs, p.values = _ptr(xs), _ptr(ys), _ptr(zs), _ptr(values)
        pr = list(params) + [0.0] * (3 - len(params))
        if int(model_id) == 0:  # linear [slope, nugget]
            pr = [params[0], params[1], 0.0]
        p.params = (C.c_double * 3)(*[float(v) for v in pr])
        p.eps, p.exact_values, p.regional_linear = float(eps), int(bool(exact_values)), int(bool(regional_linear))
        p.n_wells = 0 if wells is None else wells.shape[0]
        p.n_extra = 0 if extra_cols is None else extra_cols.shape[0]
        p.wells, p.extra_cols, p.a_inv = _ptr(wells), _ptr(extra_cols), _ptr(a_inv)
        p.geographic = int(bool(geographic))
        p.pseudo_inv = int(pseudo_inv)
        self._keep = [xs, ys, zs, values, wells, extra_cols, a_inv]
        self._taken = None
        check(self._lib.mik_set_problem(self._h, C.byref(p)))
        self._keep = []

    def assemble_only(self):
        check(self._lib.mik_assemble_only(self._h))

    def factor(self):
        check(self._lib.mik_factor(self._h))

    @property
    def order(self):
        return int(self._lib.mik_matrix_order(self._h))

    def get_matrix(self, which):
        m = self.order
        out = np.empty((m, m), dtype=np.float64)
        check(self._lib.mik_get_matrix(self._h, int(which), _ptr(out)))
        return out

    def set_points(self, px, py, pz=None, mask=None, extra_rows=None):
        g = MikPoints()
        px, py = _f64(px), _f64(py)
        pz = _f64(pz) if pz is not None else None
        if px.size != py.size or (pz is not None and pz.size != px.size):
            raise ValueError("point arrays must have the same length")
        m8 = None
        if mask is not None:
            m8 = np.ascontiguousarray(np.asarray(mask).astype(np.int8).ravel())
            if m8.size != px.size:
                raise ValueError("mask length must equal the number of points")
        er = None
        if extra_rows is not None and np.size(extra_rows):
            er = _f64(extra_rows).reshape(-1, px.size)
        g.npt = px.size
        g.px, g.py, g.pz = _ptr(px), _ptr(py), _ptr(pz)
        g.mask = m8.ctypes.data_as(C.POINTER(C.c_int8)) if m8 is not None else None
        g.extra_rows = _ptr(er)
        self._npt = px.size
        self._taken = None  # results of an earlier predict no longer belong to the resident points
        check(self._lib.mik_set_points(self._h, C.byref(g)))

    def adjust_points(self, center, rot, stretch):
        """Anisotropy adjustment of the points set_points uploaded raw, in place on the device (mik_adjust_points)."""
        c = np.zeros(3)
        c[:len(center)] = np.asarray(center, dtype=np.float64)
        r = np.zeros(9)
        r[:np.size(rot)] = np.asarray(rot, dtype=np.float64).ravel()
        st = np.ones(3)
        st[:len(stretch)] = np.asarray(stretch, dtype=np.float64)
        check(self._lib.mik_adjust_points(self._h, _ptr(c), _ptr(r), _ptr(st)))

    def set_grid(self, axes, center=None, rot=None, stretch=None, mask=None, extra_rows=None, cell_range=None):
        """The prediction points of style='grid' / 'masked' from their axes (mik_set_grid): meshgrid order and anisotropy
        adjustment on the device, H2D = the axes.  axes = (gx, gy[, gz]); center / rot (d x d) / stretch (d,) as
        core.anisotropy_matrices returns them, or all None for coordinates taken as they are (geographic).  cell_range =
        (first, count) kriges only that part of the flattened grid (mask / extra_rows / results are relative to it)."""
        g = MikGrid()
        ax = [_f64(np.asarray(a).ravel()) for a in axes]
        nd = len(ax)
        g.ndim, g.adjust = nd, int(rot is not None)
        g.nx, g.ny, g.nz = ax[0].size, ax[1].size, ax[2].size if nd == 3 else 1
        g.gx, g.gy, g.gz = _ptr(ax[0]), _ptr(ax[1]), _ptr(ax[2]) if nd == 3 else None
        if rot is not None:
            g.center = (C.c_double * 3)(*([float(v) for v in center] + [0.0] * (3 - nd)))
            g.rot = (C.c_double * 9)(*([float(v) for v in np.asarray(rot, dtype=np.float64).ravel()] + [0.0] * (9 - nd * nd)))
            g.stretch = (C.c_double * 3)(*([float(v) for v in stretch] + [1.0] * (3 - nd)))
        ncell = int(g.nx) * int(g.ny) * int(g.nz)
        first, count = (0, ncell) if cell_range is None else (int(cell_range[0]), int(cell_range[1]))
        g.cell_first, g.cell_count = (first, count) if cell_range is not None else (0, -1)  # -1 = the whole grid; 0 = an empty range
        m8 = None
        if mask is not None:
            m8 = np.ascontiguousarray(np.asarray(mask).ravel()).view(np.int8) if np.asarray(mask).dtype == np.bool_ else \
                np.ascontiguousarray(np.asarray(mask).astype(np.int8).ravel())
            if m8.size != count:
                raise ValueError("mask length must equal the number of cells")
        er = None
        if extra_rows is not None and np.size(extra_rows):
            er = _f64(extra_rows).reshape(-1, count) if count > 0 else None
        g.mask = m8.ctypes.data_as(C.POINTER(C.c_int8)) if m8 is not None else None
        g.extra_rows = _ptr(er)
        self._npt = count
        self._taken = None  # results of an earlier predict no longer belong to the resident points
        check(self._lib.mik_set_grid(self._h, C.byref(g)))

    def get_points(self, ndim):
        """Adjusted coordinates of the unmasked points resident on the device(s), (n, ndim) -- diagnostic."""
        n = int(self._lib.mik_points_resident(self._h))
        cols = [np.empty(n, dtype=np.float64) for _ in range(ndim)]
        check(self._lib.mik_get_points(self._h, _ptr(cols[0]), _ptr(cols[1]), _ptr(cols[2]) if ndim == 3 else None))
        return np.stack(cols, axis=1)

    def predict(self):
        self._taken = None  # the previous results stay valid for whoever holds them; the handle lets go of them
        check(self._lib.mik_predict(self._h))

    def predict_moving_window(self, n_closest_points):
        self._taken = None
        check(self._lib.mik_predict_moving_window(self._h, int(n_closest_points)))

    def statistics(self, n):
        k = np.zeros(n, dtype=np.float64)
        ss = np.zeros(n, dtype=np.float64)
        check(self._lib.mik_statistics(self._h, _ptr(k), _ptr(ss)))
        return k, ss

    def experimental_variogram(self, nlags):
        lags = np.zeros(int(nlags), dtype=np.float64)
        semi = np.zeros(int(nlags), dtype=np.float64)
        n = C.c_int32(0)
        check(self._lib.mik_experimental_variogram(self._h, int(nlags), _ptr(lags), _ptr(semi), C.byref(n)))
        return lags[:n.value].copy(), semi[:n.value].copy()

    def set_custom_variogram(self, gamma):
        """variogram_model='custom': `gamma(d)` maps an array of distances to semivariances (NumPy, on the host -- it is the
        user's Python code); the library hands it blocks of device-computed distances (mik_set_custom_variogram)."""
        if gamma is None:
            self._custom_cb = None
            check(self._lib.mik_set_custom_variogram(self._h, C.cast(None, VARIOGRAM_FN), None))
            return

        def _cb(_user, ptr, rows, cols, ld):
            block = np.ctypeslib.as_array(ptr, shape=(rows * ld,)).reshape(rows, ld)[:, :cols]
            block[...] = np.asarray(gamma(block), dtype=np.float64)

        self._custom_cb = VARIOGRAM_FN(_cb)  # keep it alive as long as the handle may call it
        check(self._lib.mik_set_custom_variogram(self._h, self._custom_cb, None))

    def synchronize(self):
        check(self._lib.mik_synchronize(self._h))

    def get_results(self):
        """(z, sigma^2) of the last predict, flat.  One device and no mask: the arrays ARE the page-locked landing zone the device
        wrote (mik_take_results; the buffer goes back to the library's pool when the arrays are garbage-collected) -- no copy, no
        fresh pages.  Otherwise (device groups, masked points, MIK_ZERO_COPY=0): copied / scattered into new arrays."""
        n = self._npt
        if getattr(self, "_taken", None) is not None:  # asked again before the next predict: the same arrays
            return self._taken
        if n > 0 and os.environ.get("MIK_ZERO_COPY", "1") != "0":
            zp, sp = _dp(), _dp()
            if self._lib.mik_take_results(self._h, C.byref(zp), C.byref(sp)) == MIK_OK:
                buf = (C.c_double * (2 * n)).from_address(C.addressof(zp.contents))
                buf._mik_owner = _PinnedOwner(self._lib, zp)  # released when the last view of `both` is gone
                both = np.frombuffer(buf, dtype=np.float64)
                self._taken = (both[:n], both[n:])
                return self._taken
        z = np.empty(n, dtype=np.float64)
        ss = np.empty(n, dtype=np.float64)
        check(self._lib.mik_get_results(self._h, _ptr(z), _ptr(ss)))
        return z, ss

    def timing(self):
        t = MikTiming()
        check(self._lib.mik_get_timing(self._h, C.byref(t)))
        return t.as_dict()

    # --- multi-GPU -------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        check(load().mik_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, nranks, rank, uid):
        if len(uid) != 128:
            raise ValueError("unique id must be 128 bytes")
        self.option_epoch += 1  # (a handle with a communicator is never parked)
        check(self._lib.mik_comm_init(self._h, int(nranks), int(rank), C.create_string_buffer(uid, 128)))

    def bcast_factor(self, root=0):
        check(self._lib.mik_bcast_factor(self._h, int(root)))

    def factor_checksum(self):
        """4 integers: order-independent checksums of the inverted matrix and of c as this handle's device holds them."""
        out = (C.c_uint64 * 4)()
        check(self._lib.mik_factor_checksum(self._h, out))
        return tuple(int(v) for v in out)

    def exchange_note(self):
        """Why exchange paths of the last factor() were given up ('' if none were)."""
        return (self._lib.mik_exchange_note(self._h) or b"").decode("utf-8", "replace")


def station_order(xs, ys, zs=None, geographic=False):
    """The Hilbert-curve order option "sparse" lays the stations out in (mik_station_order): order[i] = index of the station at
    position i (geographic: the same curve through (lon, lat); only the boxes of the range test are boxes of unit vectors).  Diagnostic;
    needs no GPU."""
    xs, ys = _f64(xs), _f64(ys)
    zs = _f64(zs) if zs is not None else None
    p = MikProblem()
    p.ndim, p.n = (3 if zs is not None else 2), xs.size
    p.geographic = 1 if geographic else 0
    p.xs, p.ys, p.zs = _ptr(xs), _ptr(ys), _ptr(zs)
    out = np.zeros(xs.size, dtype=np.int32)
    check(load().mik_station_order(C.byref(p), out.ctypes.data_as(C.POINTER(C.c_int32))))
    return out


def slab_of(n, members, i):
    """(lo, count) of the contiguous slab of n unmasked points that member i of a device group of `members` GPUs kriges."""
    lo, cnt = C.c_int64(0), C.c_int64(0)
    check(load().mik_slab_of(int(n), int(members), int(i), C.byref(lo), C.byref(cnt)))
    return lo.value, cnt.value


def set_devices(n):
    """Process-wide default: every handle created from now on spans `n` GPUs of the node (0 = all visible, 1 = one).  The
    environment variable MIK_NGPU does the same without touching the script."""
    flush_handle_pool()  # (parked handles were created under the old default)
    check(load().mik_set_devices(int(n)))


# ---- parked handles.  mik_create + mik_destroy cost 20 - 30 ms (four HIP streams with their queues, events, later the buffers' hipFree) where a whole
# execute() of a few hundred stations takes 1 - 3 ms: code that builds many small kriging objects (cross-validation, sliding windows) spent 95 % of its time
# there (scripts/small_object_breakdown.py).  A kriging object that goes away therefore PARKS its handle instead of destroying it, and the next object takes
# it over -- but only a handle nobody touched: no option or device group set through it, no custom variogram callback, created by this process under the same
# MIK_* environment (the library reads its option defaults from it at mik_create), and holding little device memory (the caller's estimate against
# MIK_HANDLE_POOL_BYTES, default 512 MiB).  MIK_HANDLE_POOL = how many may be parked (default 4, 0 = off).  Every call that uses a handle first replaces
# what the previous owner left in it: mik_set_problem, mik_set_points / mik_set_grid.
import threading as _threading  # noqa: E402

_pool_lock = _threading.Lock()
_pool = []


def _env_sig():
    return tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("MIK_") or k == "LOCAL_RANK"))


def acquire_handle():
    """A parked handle that fits the current process and environment, else a new one."""
    sig, pid = _env_sig(), os.getpid()
    with _pool_lock:
        for i, h in enumerate(_pool):
            if h.pid == pid and h.env_sig == sig:
                return _pool.pop(i)
    return Handle()


def release_handle(h, device_bytes):
    """Park `h` for the next kriging object, or destroy it (touched, large, pool full or switched off)."""
    if h is None or not getattr(h, "_h", None):
        return
    try:
        limit = int(os.environ.get("MIK_HANDLE_POOL", "4"))
        cap = float(os.environ.get("MIK_HANDLE_POOL_BYTES", str(512 * 2 ** 20)))
    except ValueError:
        limit, cap = 0, 0.0
    if limit > 0 and h.option_epoch == 0 and getattr(h, "_custom_cb", None) is None and h.pid == os.getpid() and device_bytes <= cap:
        with _pool_lock:
            if len(_pool) < limit:
                h._keep = []  # (the previous owner's arrays: set_problem copied them to the device long ago)
                _pool.append(h)
                return
    h.close()


def flush_handle_pool():
    with _pool_lock:
        parked, _pool[:] = list(_pool), []
    for h in parked:
        if h.pid == os.getpid():  # (a forked child must not tear down the parent's HIP objects)
            h.close()
        else:
            h._h = None


import atexit as _atexit  # noqa: E402

_atexit.register(flush_handle_pool)  # parked handles go while the HIP runtime is still there, not at module teardown


def selftest_exchange(members, init_limit_s, bcast_limit_s):
    """The RCCL path of the device-group exchange against whatever MIK_RCCL_LIB names, with stand-in members and no HIP call
    (runs without a GPU).  Returns (rc, report)."""
    buf = C.create_string_buffer(512)
    rc = load().mik_selftest_exchange(int(members), float(init_limit_s), float(bcast_limit_s), buf, 512)
    return rc, buf.value.decode("utf-8", "replace")


def selftest_mfma(device=0):
    check(load().mik_selftest_mfma(int(device)))


def selftest_exp(x, device=0):
    """The library's 19-instruction exponential (exp_neg_lean: the moving window's matrix set-up) evaluated on the device at x (<= 0)."""
    x = _f64(x).ravel()
    out = np.empty_like(x)
    check(load().mik_selftest_exp(int(device), _ptr(x), _ptr(out), int(x.size)))
    return out
