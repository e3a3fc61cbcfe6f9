"""Timeline of ONE block-sweep inverse (K2) from a rocprofv3 kernel trace: per-kernel totals, the serial panel chain of each
step (diagonal inverse -> panel kernel[s]) with the idle gaps inside it, and how much of the wall time the chain covers.

  workload : python scripts/inverse_timeline.py run [N]        (three factorisations of config-2-like stations)
  analysis : python scripts/inverse_timeline.py parse <dir with *kernel_trace.csv>
"""
import csv, glob, sys
import numpy as np


def run(n):
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import synth, internal_params
    from pykrige_amd import _lib
    c, v = synth(2, n, 2)
    h = _lib.Handle(0)
    for opt in sys.argv[3:]:
        k, val = opt.split("=")
        h.set_option(k, float(val))
    h.set_problem(ndim=2, xs=c[0], ys=c[1], zs=None, values=v, model_id=_lib.MODEL_IDS["exponential"], params=internal_params("exponential", [1.0, 0.3, 0.0]))
    for _ in range(3):
        h.factor()
    print("invert_ms", h.timing()["invert_ms"])


def parse(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void mik::", "")))
    rows.sort()
    # the last factorisation = everything from the second-to-last k_assemble on (the last one belongs to the probe of the inverse,
    # verify_inverse: assembly + k_matvec + k_matvec3, listed with the rest)
    asm = [i for i, r in enumerate(rows) if r[2].startswith("k_assemble")]
    last = asm[-2] if len(asm) >= 2 and any(r[2].replace("mik::", "").startswith("k_matvec3") for r in rows[asm[-1]:]) else asm[-1]
    rows = [r for r in rows[last:] if not r[2].startswith("k_cvec")]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    print("one factorisation: %.3f ms wall, %d kernels" % ((t1 - t0) * 1e-6, len(rows)))
    tot = {}
    for s, e, k in rows:
        a = tot.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) * 1e-3
    for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print("  %-34s x %4d   total %9.1f us   mean %7.1f us" % (k, n, us, us / n))
    # chain: from the start of each diagonal inverse to the end of the last panel kernel before the next trailing update finishes
    diag = [r for r in rows if r[2].startswith("k_diag_inv")]
    chain_k = ("k_diag_inv", "k_copy_panel", "k_panel")
    chain = [r for r in rows if r[2].replace("mik::", "").startswith(chain_k)]
    spans, gaps = [], []
    for i, d0 in enumerate(diag):
        nxt = diag[i + 1][0] if i + 1 < len(diag) else t1
        ks = [r for r in chain if d0[0] <= r[0] < nxt]
        spans.append((ks[-1][1] - ks[0][0]) * 1e-3)
        gaps.append(sum(max(0, b[0] - a[1]) for a, b in zip(ks, ks[1:])) * 1e-3)
    print("  panel chain per step: mean %.1f us (min %.1f, max %.1f), of which idle gaps between its kernels %.1f us" % (np.mean(spans), min(spans), max(spans), np.mean(gaps)))
    step = np.diff([d[0] for d in diag]) * 1e-3
    print("  step period (diag inverse start to start): mean %.1f us; chain covers %.0f %% of it" % (step.mean(), 100 * np.mean(spans[:-1]) / step.mean()))
    if len(diag) > 22:  # three steps from the middle, kernel by kernel (start relative to the first, duration; microseconds)
        w0, w1 = diag[20][0], diag[23][0] if len(diag) > 23 else t1
        print("  steps 20-22, kernel by kernel:")
        for s_, e_, k_ in rows:
            if w0 <= s_ < w1:
                print("    %8.1f  +%7.1f  %s" % ((s_ - w0) * 1e-3, (e_ - s_) * 1e-3, k_))
    upd = [r for r in rows if r[2].startswith("k_update")]
    big = [(e - s) * 1e-3 for s, e, k in upd if (e - s) > 30000]
    if big:
        print("  large k_update launches: mean %.1f us" % np.mean(big))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 5000)
    else:
        parse(sys.argv[2])
