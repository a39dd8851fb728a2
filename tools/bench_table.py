"""Markdown table of a bench.py full record (gpurun_out/bench_full.json / profiles/rNN_bench_default.json) next to an earlier
round's: python tools/bench_table.py profiles/r04_bench_default.json profiles/r03_bench_default.json"""
import json
import sys

new = json.load(open(sys.argv[1]))
old = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else {}
oldm = {m["id"]: m for m in old.get("matrix", [])}
r = new["roofline"]
print("headline: %.1f k events/s (earlier %.1f k), kernel %.1f ms, VALU %.3f, floor %.3f, HBM %.4f, cpu %.0f, dpsi p_row %s max z %s"
      % (new["value"] / 1e3, old.get("value", 0) / 1e3, r["kernel_ms"], r["frac"] or 0, r.get("floor_frac") or 0, r["hbm_measured_frac"] or 0,
         new["cpu_baseline"]["value"], new["delta_psi"]["p_row"], new["delta_psi"]["max_z"]))
print("| row | kernel(s) | events/s (earlier) | kernel ms | clock GHz | VALU busy (model, at that clock) | floor / measured | HBM | reference, 16 cores | Δψ p_row (max z) |")
print("|---|---|---|---|---|---|---|---|---|---|")
for m in new["matrix"]:
    dp = m.get("delta_psi") or {}
    o = oldm.get(m["id"], {})
    k = m["kernel"]
    k = k if len(k) < 48 else k[:45] + "…"
    print("| `%s` | `%s` | %.1f k (%.1f k) | %.1f | %s | %s | %s | %s | %.0f | %s (%.1f) |" % (
        m["id"], k, m["events_per_s"] / 1e3, o.get("events_per_s", 0) / 1e3, m["kernel_ms"],
        "–" if m.get("clock_ghz") is None else "%.2f" % m["clock_ghz"],
        "–" if m["valu_frac"] is None else "%.2f" % m["valu_frac"], "–" if m.get("floor_frac") is None else "%.2f" % m["floor_frac"],
        "–" if m["hbm_measured_frac"] is None else "%.3f" % m["hbm_measured_frac"],
        (m.get("cpu_baseline") or {}).get("value", 0), dp.get("p_row"), dp.get("max_z", 0)))
