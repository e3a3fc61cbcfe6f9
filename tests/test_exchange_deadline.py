"""The factor exchange of a device group cannot hang (include/mikrige.h, "Bounded waits"): every RCCL call of it runs on a
worker thread under a limit, a stuck call is abandoned and the exchange goes on over peer copies.

CPU part: mik_selftest_exchange drives the RCCL path of the exchange -- communicator set-up, the grouped broadcast, the
wait with its two limits -- with stand-in members and without a single HIP call, against a stand-in librccl
(tests/standin/standin_rccl.cpp via MIK_RCCL_LIB) whose calls succeed, fail or NEVER RETURN.
GPU part: the same stand-ins under a real mik_factor on a device group (aliased onto the one GPU): the call returns within
the limit through the peer path, says so, and the results are bit-identical to one device's."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _standins():
    from tests.standin import build

    return build.build()


def _selftest(mode, members=4, init=0.6, bcast=0.6):
    cpu, _ = _standins()
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from pykrige_amd import _lib\n"
            "import time; t0 = time.time()\n"
            "rc, rep = _lib.selftest_exchange(%d, %r, %r)\n"
            "rc2, rep2 = _lib.selftest_exchange(%d, %r, %r)\n"  # a second exchange in the same process
            "print('%%d|%%s|%%d|%%s|%%.2f' %% (rc, rep, rc2, rep2, time.time() - t0))\n" % (ROOT, members, init, bcast, members, init, bcast))
    env = dict(os.environ, MIK_RCCL_LIB=cpu, STANDIN_RCCL_MODE=mode)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-800:]
    rc, rep, rc2, rep2, dt = r.stdout.strip().splitlines()[-1].split("|")
    return int(rc), rep, int(rc2), rep2, float(dt), time.time() - t0


def _timeout_report(rep, phase, limit):
    """'timeout phase=<phase> after <t> s; ...': the wait ran out in that phase, no earlier than its limit and (on a loaded test host
    the waiting thread may be scheduled late) well before the OTHER limit or any multiple of this one could have expired."""
    import re

    m = re.match(r"timeout phase=%s after ([0-9.]+) s" % phase, rep)
    assert m, rep
    t = float(m.group(1))
    assert limit - 0.01 <= t <= limit + 1.5, rep
    return True


def test_exchange_with_a_working_rccl_reports_its_ranks():
    rc, rep, rc2, rep2, dt, _ = _selftest("ok", members=8)
    assert rc == 0 and rep == "ok ranks=8" and rc2 == 0 and rep2 == "ok ranks=8"  # communicators are cached: the second one too
    assert dt < 5.0


def test_comm_init_that_never_returns_is_abandoned_within_the_limit():
    rc, rep, rc2, rep2, dt, wall = _selftest("hang_init")
    assert rc == -4 and _timeout_report(rep, "init", 0.6) and "rccl_dead=1" in rep, rep
    # RCCL is not tried again in this process: the second exchange fails at once instead of waiting out another limit
    assert rc2 == -4 and "RCCL disabled for this process" in rep2 and "ncclCommInitAll" in rep2, rep2
    assert dt < 6.0 and wall < 60.0  # and the process exits although a worker thread is still stuck inside the stand-in


def test_broadcast_that_never_returns_is_abandoned_within_the_limit():
    rc, rep, rc2, rep2, dt, _ = _selftest("hang_bcast", init=5.0, bcast=0.5)
    assert rc == -4 and _timeout_report(rep, "bcast", 0.5), rep  # the init limit (5 s) was NOT what expired
    assert rc2 == -4 and "RCCL disabled" in rep2 and "ncclBroadcast" in rep2
    assert dt < 6.0


def test_comm_init_that_fails_is_an_error_not_a_wait():
    rc, rep, rc2, rep2, dt, _ = _selftest("fail_init", init=30.0, bcast=30.0)
    assert rc == -4 and rep.startswith("failed:") and "CommInitAll" in rep
    assert rc2 == -4 and rep2.startswith("failed:")  # a failure (not a stall) does not disable RCCL: it is simply tried again
    assert dt < 6.0


# ------------------------------------------------------------------------------------------------ on the GPU
_GPU_SCRIPT = r"""
import sys, time, json
sys.path.insert(0, %(root)r)
import numpy as np
from pykrige_amd import _lib
from tests import _fixtures as fx
c, v = fx.synth(11, 700, 2)
rng = np.random.default_rng(3)
pts = [rng.random(5000), rng.random(5000)]
def run(h):
    h.set_problem(ndim=2, xs=c[0], ys=c[1], zs=None, values=v, model_id=_lib.MODEL_IDS["exponential"], params=[0.9, 0.3, 0.1])
    h.set_points(pts[0], pts[1])
    t0 = time.time(); h.factor(); tf = time.time() - t0
    h.predict()
    return h.get_results(), tf
h1 = _lib.Handle(0)
(z1, s1), _ = run(h1)
hg = _lib.Handle(0)
hg.set_devices(3, alias=True)
hg.set_option("exchange", %(exchange)d)
out = {}
try:
    (zg, sg), tf = run(hg)
    t = hg.timing()
    out = dict(ok=True, same=bool(np.array_equal(z1, zg) and np.array_equal(s1, sg)), path=t["exchange_path"], fallbacks=t["exchange_fallbacks"],
               ranks=t["rccl_ranks"], note=hg.exchange_note(), factor_s=tf, wait_ms=t["exchange_wait_ms"], ms=t["exchange_ms"])
    t0 = time.time()
    (zg2, sg2), tf2 = run(hg)   # the next factor on the same handle (and, after a stall, with RCCL disabled)
    t2 = hg.timing()
    out.update(same2=bool(np.array_equal(z1, zg2)), path2=t2["exchange_path"], second_s=time.time() - t0, note2=hg.exchange_note())
except Exception as e:
    out = dict(ok=False, err=repr(e))
print("RESULT " + json.dumps(out))
"""


def _gpu_case(mode, exchange=0, init="1.0", bcast="1.0"):
    import json

    _, hip = _standins()
    assert os.path.exists(hip), "the HIP stand-in needs hipcc"
    env = dict(os.environ, MIK_RCCL_LIB=hip, STANDIN_RCCL_MODE=mode, MIK_RCCL_ALLOW_ALIAS="1", MIK_RCCL_INIT_TIMEOUT=init,
               MIK_RCCL_BCAST_TIMEOUT=bcast)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", _GPU_SCRIPT % dict(root=ROOT, exchange=exchange)], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-800:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[7:]), time.time() - t0


@pytest.mark.gpu
def test_group_exchange_over_a_working_rccl_standin():
    o, _ = _gpu_case("ok")
    assert o["ok"] and o["same"] and o["path"] == 1 and o["ranks"] == 3 and o["fallbacks"] == 0 and o["note"] == "", o
    assert o["same2"] and o["path2"] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("mode,what", [("hang_init", "ncclCommInitAll did not finish within 1.0 s"),
                                       ("hang_bcast", "the grouped ncclBroadcast did not finish within 1.0 s"),
                                       ("corrupt", "checksum mismatch on group member 1"),
                                       ("fail_init", "stand-in RCCL error")])
def test_group_exchange_survives_a_stuck_or_broken_rccl(mode, what):
    """exchange = auto: mik_factor + mik_predict return through the peer path within the limit, bit-identical results."""
    o, wall = _gpu_case(mode)
    assert o["ok"], o
    assert o["same"] and o["path"] == 2 and o["fallbacks"] == 1 and what in o["note"] and "rccl broadcast failed" in o["note"], o
    assert o["factor_s"] < 1.0  # asynchronous exchange: mik_factor itself returned at once
    assert o["same2"] and o["path2"] == 2, o
    if mode.startswith("hang"):
        assert "rccl disabled for this process" in o["note2"] and o["second_s"] < 0.9, o  # no second wait on a dead RCCL
    assert wall < 120.0


@pytest.mark.gpu
def test_forced_rccl_that_hangs_is_an_error_not_a_hang():
    o, wall = _gpu_case("hang_bcast", exchange=1)
    assert not o["ok"] and "RuntimeError" in o["err"] and "did not finish within" in o["err"], o
    assert wall < 120.0


@pytest.mark.gpu
def test_bench_with_eight_members_survives_a_hanging_rccl_within_its_budget():
    """The first real 8-GPU run, rehearsed: `bench.py --gpus 8 --config 5` as the SCALE driver starts it (eight members, here aliased
    onto the one GPU) against an RCCL whose ncclCommInitAll never returns.  One valid JSON line must come out, the exchange that
    ran is named with the reason the default one was given up, and the trials before the timed loop respect their wall-clock
    budget (--pretrial-budget; --no-trials skips them)."""
    import json

    _, hip = _standins()
    assert os.path.exists(hip), "the HIP stand-in needs hipcc"
    env = dict(os.environ, MIK_RCCL_LIB=hip, STANDIN_RCCL_MODE="hang_init", MIK_RCCL_ALLOW_ALIAS="1", MIK_RCCL_INIT_TIMEOUT="2.0",
               MIK_RCCL_BCAST_TIMEOUT="2.0", MIK_PEER_TIMEOUT="30")
    for extra, budget in ((["--pretrial-budget", "5"], 5.0), (["--no-trials"], None)):
        t0 = time.time()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "5", "--steps", "1", "--warmup", "0",
                            "--no-cpu"] + extra, capture_output=True, text=True, timeout=600, env=env)
        wall = time.time() - t0
        assert r.returncode == 0, r.stderr[-1500:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-500:]
        o = json.loads(lines[0])
        assert o["value"] and o["n_gpus"] == 8 and o["config"]["grid_points_total"] == 8 * 4096 * 512
        fx_ = o["config"]["factor_exchange"]
        assert fx_.startswith("peer_scatter_allgather") and ("did not finish within" in fx_ or "rccl disabled" in fx_), fx_
        tr = o["factor_exchange_trial"]  # (top level; the driver's record keeps the flat scalar copies config["trial_*"])
        if budget is None:
            assert "skipped" in tr and "trial_skipped" in o["config"]
        else:
            # a trial in flight when the budget runs out finishes (its own waits are bounded by the library's limits); nothing new starts
            assert tr["budget_s"] == budget and tr["spent_s"] < budget + 45.0, tr
            assert o["config"]["trial_budget_s"] == budget
        # what an 8-GPU record must show: which exchange ran, on how many RCCL ranks, what it moved (the packed upper triangle + c)
        cfg = o["config"]
        assert all(v is None or isinstance(v, (int, float, str, bool)) for v in cfg.values()), cfg
        assert cfg["exchange_path"].startswith("peer") and cfg["rccl_ranks"] == 0 and cfg["exchange_bytes"] == 8.0 * (8064 * (8064 + 128) // 2 + 8064), cfg
        assert cfg["predict_ms_slowest_device"] >= cfg["predict_ms_fastest_device"] > 0.0
        assert wall < 300.0, wall


_RANKS_SCRIPT = r"""
import sys, json
sys.path.insert(0, %(root)r)
import numpy as np
from pykrige_amd import _lib
from tests import _fixtures as fx
c, v = fx.synth(11, %(n)d, 2)
rng = np.random.default_rng(3)
pts = [rng.random(4000), rng.random(4000)]
uid = _lib.Handle.comm_unique_id()
hs = []
for rank in range(2):  # two "ranks" in one process (the stand-in hands the root's buffer to the member: the root must call first)
    h = _lib.Handle(0)
    h.set_option("exchange_tri", %(tri)d)
    h.set_problem(ndim=2, xs=c[0], ys=c[1], zs=None, values=v, model_id=_lib.MODEL_IDS[%(model)r], params=%(params)r)
    h.comm_init(2, rank, uid)
    hs.append(h)
hs[0].factor()
for h in hs:
    h.bcast_factor(0)
sums = [list(h.factor_checksum()) for h in hs]
res = []
for h in hs:
    h.set_points(pts[0], pts[1])
    h.predict()
    res.append(h.get_results())
a0, a1 = hs[0].get_matrix(1), hs[1].get_matrix(1)  # the member mirrored the triangle it received: the root's matrix, bit for bit
print("RESULT " + json.dumps(dict(sums_equal=sums[0] == sums[1], same=bool(np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])),
                                  bytes=[h.timing()["exchange_bytes"] for h in hs], matrix_gap=float(np.abs(a0 - a1).max() / np.abs(a0).max()),
                                  symmetric=bool(np.array_equal(a1, a1.T)))))
"""


@pytest.mark.gpu
@pytest.mark.parametrize("model,params,n", [("exponential", [0.9, 0.3, 0.1], 1300), ("spherical", [0.95, 0.25, 0.05], 3300)])
def test_rank_broadcast_moves_the_upper_block_triangle(model, params, n):
    """One process per GPU (mik_comm_init + mik_bcast_factor, what `torch.distributed.run bench.py` drives): round 6 broadcasts the packed upper
    block triangle.  Two ranks in one process over the stand-in RCCL: the ranks' checksums (of the packed triangle) agree, the non-root
    rank unpacks and mirrors it and kriges bit-identically (dense contraction at 11 block columns, range-aware and the half sweep at 26),
    its matrix is the root's bit for bit, and the bytes are those of the triangle; "exchange_tri" 0 moves the square."""
    import json

    _, hip = _standins()
    mp = 128 * ((n + 1 + 127) // 128)
    for tri in (1, 0):
        env = dict(os.environ, MIK_RCCL_LIB=hip, STANDIN_RCCL_MODE="ok")
        r = subprocess.run([sys.executable, "-c", _RANKS_SCRIPT % dict(root=ROOT, n=n, tri=tri, model=model, params=params)], capture_output=True,
                           text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-800:]
        o = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
        assert o["sums_equal"] and o["same"], o
        want = 8.0 * ((mp * (mp + 128) // 2 if tri else mp * mp) + mp)
        assert o["bytes"] == [want, want], (o, want)
        assert o["matrix_gap"] == 0.0 and o["symmetric"], o
