"""Per-kernel average L2-miss latency (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ) from the counter CSVs scripts/gpu_hbm_split.sh collected, and
the split of k_contract's fabric reads into Infinity-Cache hits and HBM reads it implies:
    L(kernel) = f L_hbm + (1 - f) L_cache   with L_hbm, L_cache from tools/ea_probe's two calibration kernels under the same counters.
A latency-based estimate (queueing under load moves both ends); it bounds the HBM side where no hit / miss counter exists."""
import csv, glob, json, os, sys

out = sys.argv[1]


def per_kernel(sub):
    rows = {}
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            d = rows.setdefault(k, {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            d["_n_" + r["Counter_Name"]] = d.get("_n_" + r["Counter_Name"], 0) + 1
    return rows


probe, bench, fetch = per_kernel("probe"), per_kernel("bench"), per_kernel("fetch")


def lat(d):
    return d["TCC_EA0_RDREQ_LEVEL_sum"] / max(1.0, d["TCC_EA0_RDREQ_sum"])


cal = {}
for k, d in probe.items():
    if "TCC_EA0_RDREQ_sum" in d:
        cal[k] = lat(d)
        print("calibration %-22s requests %.4g  of them 'DRAM' %.4g  32-byte %.4g  average latency %.0f TCC cycles" % (
            k, d["TCC_EA0_RDREQ_sum"], d.get("TCC_EA0_RDREQ_DRAM_sum", float("nan")), d.get("TCC_EA0_RDREQ_32B_sum", float("nan")), cal[k]))
l_hbm_loaded = next((v for k, v in cal.items() if "hbm" in k), None)
l_mall_loaded = next((v for k, v in cal.items() if "mall" in k), None)
print("(the two calibration kernels saturate the fabric -- 5.0-5.7 TB/s -- so their latencies include queueing; the end points used below come"
      " from kernels of the SAME bench run whose data location is known and whose request rate is moderate, like the contraction's)")
print()


def wavg(names):
    num = den = 0.0
    for k, d in bench.items():
        if any(n in k for n in names) and d.get("TCC_EA0_RDREQ_sum", 0.0) > 0:
            num += d["TCC_EA0_RDREQ_LEVEL_sum"]
            den += d["TCC_EA0_RDREQ_sum"]
    return num / den if den else None


# Infinity-Cache end point: the sweep's chain kernels re-read panels the previous kernel wrote a few microseconds earlier (a few MB)
l_cache = wavg(["k_panel", "k_gemm128", "k_diag_inv_b"])
# HBM end point, lower bound: k_ss_reduce / k_matvec3 read data that k_rhs's 5 GB of writes have pushed out (partly); upper: the streamed probe
l_hbm_lo = wavg(["k_ss_reduce", "k_matvec3"])
l_hbm_hi = l_hbm_loaded
res = {}
for k, d in sorted(bench.items(), key=lambda kv: -kv[1].get("TCC_EA0_RDREQ_sum", 0.0)):
    if d.get("TCC_EA0_RDREQ_sum", 0.0) < 1e5:
        continue
    n = d["_n_TCC_EA0_RDREQ_sum"]
    L = lat(d)
    fs = fetch.get(k, {}).get("FETCH_SIZE")
    fab = None if fs is None else 2.0 * 1024.0 * fs / max(1, fetch[k]["_n_FETCH_SIZE"])
    print("%-60s launches %4d  requests per launch %.4g  average L2-miss latency %5.0f cycles  fabric reads per launch %s" % (
        k[:60], n, d["TCC_EA0_RDREQ_sum"] / n, L, "n/a" if fab is None else "%.3g GB" % (fab / 1e9)))
    res[k] = {"launches": n, "avg_l2_miss_latency_cycles": L, "fabric_read_bytes_per_launch": fab}
kc = next((k for k in res if "k_contract" in k), None)
if kc and l_cache and l_hbm_lo and l_hbm_hi and res[kc]["fabric_read_bytes_per_launch"]:
    r = res[kc]
    L = r["avg_l2_miss_latency_cycles"]
    f_hi = min(1.0, max(0.0, (L - l_cache) / (l_hbm_lo - l_cache)))
    f_lo = min(1.0, max(0.0, (L - l_cache) / (l_hbm_hi - l_cache)))
    fab = r["fabric_read_bytes_per_launch"]
    print()
    print("end points: Infinity Cache %.0f cycles (k_panel / k_gemm128 / k_diag_inv_b of the same run), HBM %.0f (k_ss_reduce / k_matvec3) .. %.0f (6 GiB streamed)"
          % (l_cache, l_hbm_lo, l_hbm_hi))
    print("k_contract: average L2-miss latency %.0f cycles -> %.0f .. %.0f %% of its %.1f GB of fabric reads per launch come from HBM = %.1f .. %.1f GB;"
          " the rest are Infinity-Cache hits" % (L, 100 * f_lo, 100 * f_hi, fab / 1e9, f_lo * fab / 1e9, f_hi * fab / 1e9))
    print("JSON " + json.dumps({"kernel": kc, "traffic_hbm_bytes_per_launch_by_latency": [f_lo * fab, f_hi * fab], "hbm_fraction_by_latency": [f_lo, f_hi],
                                "fabric_read_bytes_per_launch": fab, "latency_cycles": {"kernel": L, "infinity_cache": l_cache, "hbm_low": l_hbm_lo, "hbm_streamed": l_hbm_hi,
                                                                                         "infinity_cache_saturated": l_mall_loaded}}))
