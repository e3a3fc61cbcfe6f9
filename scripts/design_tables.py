"""Markdown rows for DESIGN.md section 5 from the bench JSONs of an evidence run (profiles/rNN_bench_*.json)."""
import json, sys

R = sys.argv[1] if len(sys.argv) > 1 else "r03"
P = "profiles/%s_bench_" % R
names = {2: "2: OK2D N=5000, 1000×1000, exponential", 3: "3: OK3D N=2000, 200×200×50, gaussian",
         4: "4: UK2D N=4000, 1024², regional_linear + 3 wells", 5: "5: OK2D N=8000, 4096×512 slab (1/8 of 4096²)"}
print("| config | `value`: execute() points/s | ms per execute() (assemble / invert / rhs / contract) | resident step ms (execute − resident) | K3b executed TFLOP/s (`frac`) / effective | fabric bytes per launch (live PMC) | `style='points'` points/s |")
print("|---|---|---|---|---|---|---|")
for c in (2, 3, 4, 5):
    d = json.load(open(P + "c%d.json" % c))
    p, r = d["phases_ms_per_step"], d["roofline"]
    print("| %s | **%.2f M** | %.1f (%.2f / %.2f / %.1f / %.1f) | %.1f (%+.2f ms = %.2f %%) | %.1f (%.3f) / %.1f | %.0f GB (algorithmic %.1f) | %.2f M (%.1f %%) |" % (
        names[c], d["value"] / 1e6, d["ms_per_step"], p["assemble"], p["invert"], p["rhs"], p["contract"], d["resident"]["ms_per_step"],
        d["host_overhead"]["ms_per_step"], 100 * d["host_overhead"]["frac"], r["achieved"], r["frac"], r["effective_tflops"],
        (r["traffic"] or 0) / 1e9, r["algorithmic_bytes_per_launch"] / 1e9, d["execute_points_style"]["value"] / 1e6,
        100 * (d["execute_points_style"]["value"] / d["value"] - 1)))
print()
print("| config | CPU leg | kind | best threads | slab points | points/s incl. fixed costs | steady state | GPU ÷ steady state | GPU vs CPU max\\|Δz\\| / max\\|Δσ²\\| |")
print("|---|---|---|---|---|---|---|---|---|")
for c in (2, 3, 4, 5):
    d = json.load(open(P + "c%d.json" % c))
    cb = d["cpu_baseline"]
    what = "the reference's own `backend='C'` loop (`lib/cok.pyx`, `oracle/_ref`)" if cb["kind"] == "reference" else "NumPy/SciPy restatement of `backend='vectorized'`"
    print("| %d | %s | %s | %s | %d | %.0f | %.0f | %.0f | %.1e / %.1e |" % (c, what, cb["kind"], cb["cores"], cb["slab_points"], cb["value"], cb["steady_state"],
                                                                    d["value"] / cb["steady_state"], cb["gpu_vs_cpu_max_abs_dz"], cb["gpu_vs_cpu_max_abs_dss"]))
    if cb["kind"] == "reference" and "vectorized" in cb:
        v = cb["vectorized"]
        print("| %d | NumPy/SciPy restatement of `backend='vectorized'` | port | %s | %d | %.0f | %.0f | %.0f | |" % (c, v["cores"], cb["slab_points"], v["value"], v["steady_state"], d["value"] / v["steady_state"]))
print()
print("| moving window | points/s | ms per call | solver kernel | achieved / frac | traffic per launch | CPU (reference `_c_exec_loop_moving_window` + cKDTree) |")
print("|---|---|---|---|---|---|---|")
for k in (10, 50, 100):
    d = json.load(open(P + "moving_window_k%d.json" % k))
    r, cb = d["roofline"], d["cpu_baseline"]
    print("| k = %d | %.1f M | %.2f | `%s` | %.2f TFLOP/s / %.3f | %s | %.0f points/s at %s threads |" % (k, d["value"] / 1e6, d["ms_per_step"], r["kernel"], r["achieved"], r["frac"],
                                                                                        ("%.2f GB" % (r["traffic"] / 1e9)) if r.get("traffic") else "—", cb["value"], cb["cores"]))
print()
for n, f in (("device group of 2", "c2_group2_aliased_on_1gpu"), ("device group of 4", "c2_group4_aliased_on_1gpu"), ("device group of 8", "c2_group8_aliased_on_1gpu"),
             ("device group of 8, config 5", "c5_group8_aliased_on_1gpu"), ("torchrun, 2 ranks", "torchrun_2ranks_on_1gpu")):
    d = json.load(open(P + f + ".json"))
    ho = d.get("host_overhead", {})
    print("%s: %.3f M points/s, %.1f ms per execute(), resident %s ms, execute − resident %s ms (%s %%), exchange: %s" % (
        n, d["value"] / 1e6, d["ms_per_step"], ("%.1f" % d["resident"]["ms_per_step"]) if "resident" in d else "—",
        ("%.2f" % ho["ms_per_step"]) if "ms_per_step" in ho else "—", ("%.2f" % (100 * ho["frac"])) if "frac" in ho else "—", d["config"].get("factor_exchange", "")[:60]))
