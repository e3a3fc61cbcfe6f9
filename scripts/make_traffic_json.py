"""profiles/k_contract_traffic.json from a tracked per-kernel PMC summary (scripts/pmc_summary.py output) and the bench line of
the same evidence run: HBM-side bytes per launch of the dominant kernel (gfx950: read bytes = 2 x FETCH_SIZE x 1024), L2 hit rate,
MFMA-busy fraction.  bench.py falls back to this file when its live PMC passes fail."""
import json, sys

pmc, bench = sys.argv[1], sys.argv[2]
rows = {}
for line in list(open(pmc))[1:]:  # kernel names contain commas (template arguments): the four numeric fields are the LAST four
    f = line.rstrip("\n").rsplit(",", 4)
    if f[0].startswith("void mik::k_contract<"):
        rows.setdefault(f[0], {})[f[1]] = float(f[3])
kern = max(rows, key=lambda k: rows[k].get("FETCH_SIZE", 0.0))
c = rows[kern]
b = json.load(open(bench))
pts = b["config"]["grid_points_per_gpu"] / b["roofline"]["launches_per_step"]
out = {
    "kernel": kern.replace("void mik::", ""),
    "workload": b["config"]["workload"],
    "points_per_launch": pts,
    "FETCH_SIZE_KB_per_launch": c["FETCH_SIZE"],
    "WRITE_SIZE_KB_per_launch": c["WRITE_SIZE"],
    "correction": "gfx950 FETCH_SIZE counts 128-B requests as 64 B (MI355X_MICROARCH.md, HBM section): read bytes = 2 x FETCH_SIZE x 1024; calibrated in round 1 on k_cvec (reads 200.0 MB, FETCH_SIZE reports 100.3 MB) and k_rhs (writes 5.37 GB, WRITE_SIZE reports 5.12e6 KB)",
    "hbm_bytes_per_launch": 2.0 * c["FETCH_SIZE"] * 1024.0 + c["WRITE_SIZE"] * 1024.0,
    "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"],
    "tcc_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
    "mfma_busy_fraction": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0),
    "source": "%s (rocprofv3 --kernel-trace --pmc, one counter group per pass: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum ... | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ...; default event-ordered schedule of the block sweep)" % pmc,
    "note": "averages over the launches of the run (equal launches of %d points); bench.py collects FETCH_SIZE / WRITE_SIZE live and uses this file only when that fails" % int(pts),
}
print(json.dumps(out, indent=1))
