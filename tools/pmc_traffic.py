#!/usr/bin/env python3
"""Build profiles/rNN_pmc_traffic.json from two rocprofv3 --pmc passes of the bench command
(separate passes -- FETCH_SIZE; WRITE_SIZE; TCC_HIT_sum TCC_MISS_sum -- the first two do not fit one pass; --output-format csv):
    python tools/pmc_traffic.py <dir with all passes' CSVs> <out.json> "<workload description>"
HBM-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports half of wide coalesced reads
(calibrated on layernorm_kernel, whose algorithmic traffic is known exactly; the calibration row is written to the JSON)."""
import csv, glob, json, sys, collections

d1, out, desc = sys.argv[1:4]
# optional: the bench.py selection this file belongs to (bench.py looks the traffic up by these keys)
bench_workload, precision, batch = (sys.argv[4:7] + ["vitb518", "bf16", "64"][len(sys.argv[4:7]):])


def load(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:70] + " grid=" + r.get("Grid_Size", "?") + " wg=" + r.get("Workgroup_Size", "?")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


a = load(d1)
b = a
per = {}
for k, d in a.items():
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d or k.startswith("void at::") or "elementwise" in k:
        continue
    f, w = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    e = {"launches_profiled": len(d["FETCH_SIZE"]), "hbm_bytes_per_launch": (2 * f + w) * 1024, "fetch_kb_raw": round(f, 1), "write_kb": round(w, 1)}
    if k in b and "TCC_HIT_sum" in b[k]:
        h, m = sum(b[k]["TCC_HIT_sum"]), sum(b[k]["TCC_MISS_sum"])
        e["l2_hit_rate"] = h / max(1.0, h + m)
    per[k] = e
gemm = {k: v for k, v in per.items() if k.startswith("gemm_bf16") or k.startswith("void gemm_bf16") or "gemm_x3" in k or "gemm_pp" in k or "gemm_h2" in k or "patch_embed" in k}
n = sum(v["launches_profiled"] for v in gemm.values())
avg = sum(v["hbm_bytes_per_launch"] * v["launches_profiled"] for v in gemm.values()) / max(1, n)
json.dump({"workload": bench_workload, "precision": precision, "batch": int(batch), "command": desc, "source": "rocprofv3 --pmc FETCH_SIZE WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum (separate passes)",
           "correction": "HBM-side bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads; see the layernorm_kernel rows: 269.4 MB algorithmic read + 134.7 MB write per launch at 87680 x 768)",
           "gemm_bf16_avg_bytes_per_launch": avg, "gemm_launches_profiled": n, "per_kernel": dict(sorted(per.items()))}, open(out, "w"), indent=1)
print(f"{len(per)} kernels, bf16 GEMM average {avg / 1e6:.1f} MB per launch over {n} launches -> {out}")
