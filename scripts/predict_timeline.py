"""Timeline of ONE mik_predict of the range-aware contraction (BASELINE config 5) from a rocprofv3 kernel trace: which kernel ran when on
which HSA queue (= stream), how long each waited behind the one before it on its queue, and where the prediction's wall time goes that is
not contraction (round-5 review: predict_total 40.6 ms against 36.2 ms of k_contract_spg; memsets at 0.5 ms, k_ss_reduce_sp at 1 ms).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --config 5 --steps 2 --warmup 1 --no-cpu --pmc off --no-other
    python scripts/predict_timeline.py DIR [full]
"""
import csv
import glob
import sys


def short(name):
    n = name.split("(")[0].replace("void mik::", "").replace("mik::", "")
    return n if len(n) <= 44 else n[:41] + "..."


def main(d, full=False):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
    rows.sort()
    # the last prediction: from the first range-aware kernel after the last k_cvec (end of the last mik_factor) to the end of the trace
    cv = [i for i, r in enumerate(rows) if r[2].startswith("k_cvec")]
    rows = rows[cv[-1] + 1:] if cv else rows
    first = next(i for i, r in enumerate(rows) if r[2].startswith(("k_sp_cand", "k_ps_", "k_rhs", "k_geo_unit_p")))
    rows = rows[first:]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    print("one prediction: %.3f ms from the first kernel's start to the last one's end, %d dispatches on %d queues" % (
        (t1 - t0) * 1e-6, len(rows), len({r[3] for r in rows})))
    tot = {}
    for s, e, k, q in rows:
        a = tot.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += (e - s) * 1e-3
        a[2] = max(a[2], (e - s) * 1e-3)
    for k, (n, us, mx) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print("  %-46s x %4d   total %9.1f us   mean %8.1f us   max %8.1f us" % (k, n, us, us / n, mx))
    # union of the intervals in which a contraction kernel runs, and what the rest of the wall time is covered by
    con = sorted((s, e) for s, e, k, q in rows if k.startswith("k_contract"))
    cover, cur = 0, None
    for s, e in con:
        if cur is None or s > cur[1]:
            if cur:
                cover += cur[1] - cur[0]
            cur = [s, e]
        else:
            cur[1] = max(cur[1], e)
    cover += cur[1] - cur[0] if cur else 0
    print("  a contraction kernel is running for %.3f ms of the %.3f ms (sum of its launches %.3f ms: they overlap across the two lanes)" % (
        cover * 1e-6, (t1 - t0) * 1e-6, sum(e - s for s, e in con) * 1e-6))
    print("  before the first contraction starts: %.3f ms; after the last one ends: %.3f ms" % ((con[0][0] - t0) * 1e-6, (t1 - max(e for _, e in con)) * 1e-6))
    # per queue: the dependency chain of a lane -- every kernel's wait behind its predecessor on the same queue
    for q in sorted({r[3] for r in rows}):
        seq = [r for r in rows if r[3] == q]
        busy = sum(e - s for s, e, _, _ in seq)
        gaps = [(b[0] - a[1]) * 1e-3 for a, b in zip(seq, seq[1:])]
        print("  queue %s: %d dispatches, busy %.3f ms, idle between its dispatches %.3f ms (largest gap %.1f us)" % (
            q, len(seq), busy * 1e-6, sum(g for g in gaps if g > 0) * 1e-3, max(gaps) if gaps else 0.0))
    if full:
        print("  every dispatch (start us, duration us, queue, kernel):")
        for s, e, k, q in rows:
            print("    %9.1f  +%8.1f  q%-3s %s" % ((s - t0) * 1e-3, (e - s) * 1e-3, q, k))


if __name__ == "__main__":
    main(sys.argv[1], len(sys.argv) > 2)
