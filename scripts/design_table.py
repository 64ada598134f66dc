"""The measurement table of DESIGN.md section 6 from a bench.py --out file:  python scripts/design_table.py <jsonl> [<sharded-at-1 json>]"""
import json
import sys

rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
print("| line | dtype | Gsamp/s | ms/pass | kernel ms | frac of 8 TB/s | alg. B/sample | PMC traffic ÷ alg. |")
print("|---|---|---|---|---|---|---|---|")
for d in rows + ([json.loads(open(sys.argv[2]).read())] if len(sys.argv) > 2 else []):
    r, c = d["roofline"], d["config"]
    name = c["workload"].split(":")[0] if not c["workload"].startswith(("next", "api", "cfg5")) else c["workload"].split(": ")[0]
    what = c["workload"].split(", ", 1)[1] if ", " in c["workload"] else ""
    what = what.replace("compute_Sv->remove_background_noise(20x50,3dB)->compute_MVBS(20s x 1m) in two sweeps", "chain, two sweeps")
    what = what.replace("fused compute_Sv->compute_MVBS(20s x 1m)", "fused kernel").replace(" Sv dataset resident in HBM,", "")
    if "sound_speed_changes_every_n_pings" in c and c["sound_speed_changes_every_n_pings"] != 1:
        what += f", ss every {c['sound_speed_changes_every_n_pings']}"
    if "tile_streams" in c:
        what = f"{c['tile_streams']} streams; {c['route']}"
    elif name.startswith("cfg5"):
        what = "one stream; " + c["route"]
    algo = r["bytes_per_sample"] * c["samples_per_step"] / c["passes_per_step"] / (len(c["tiles"].split(" x ")[0]) and int(c["tiles"].split(" x ")[0]) if "tiles" in c else 1)
    tr = "—" if not r.get("traffic") else "%.3f" % (r["traffic"] / algo)
    each = f" ({r['kernel_ms_each']:.1f} each)" if "kernel_ms_each" in r else ""
    print(f"| `{name}` {what[:72]} | {d['dtype']} | {d['value'] / 1e9:.1f} | {c['ms_per_pass']:.2f} | {r['kernel_ms']:.2f}{each} | "
          f"**{r['frac']:.3f}** | {r['bytes_per_sample']} | {tr} |")
cb = rows[-1].get("cpu_baseline") or {}
if cb:
    print(f"| NumPy oracle on the host (`cpu_baseline`) | f64 | {cb['value'] / 1e9:.3f} on one core; "
          f"{cb.get('multicore_value', 0) / 1e9:.3f} with {cb.get('multicore_cores')} processes | | | | | |")
