"""PMC FETCH_SIZE / WRITE_SIZE passes of bench.py -> profiles/hbm_traffic.json (bytes per launch of the dominant
kernel of each workload, with the hash of the kernel sources they were measured on; bench.py reads it back into
roofline.traffic and drops it when the sources changed).
    python scripts/make_traffic_json.py <dir holding fetch_<wl>/ and write_<wl>/ rocprofv3 outputs>
Counters are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE x 2 for wide coalesced reads;
WRITE_SIZE x 1.  Calibration on epa_power_coef_ek's known traffic is recorded next to the numbers."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_hash  # noqa: E402

src = sys.argv[1]
# traffic key of bench.py -> (rocprof output tag, kernel-name substrings, must also contain, algorithmic bytes per launch)
KERNELS = {
    # (the <.., true, true> instantiation with the range-maximum by-product belongs to the Dataset-API timing of the run)
    "cfg2:float64": ("cfg2", ["fused_sv_mvbs_kernel<double, float, true, false, 0>"], None, 4 * 500_000 * 2000 * 12),
    "cfg3:float64": ("cfg3", ["sv_noise_fast_kernel", "sv_denoise_mvbs_fast_kernel", "sv_denoise_mvbs_uniform_kernel",
                              "sv_denoise_mvbs_drift_kernel"], None, 4 * 500_000 * 2000 * 32),
    "cfg4:float64": ("cfg4", ["sv_complex_fft_kernel<float, double, double"], None, 2 * 200_000 * 8192 * 40),
    "cfg4:float64:planes64": ("cfg4_planes64", ["sv_complex_fft_kernel<double, double, double"], None, 2 * 200_000 * 8192 * 72),
    "cfg5:float64": ("cfg5", ["fused_sv_mvbs_kernel<double, float, true, false, 0>"], None, 4 * 250_000 * 4096 * 12),
    # the round-4 headline: the same tiles through calibrate.compute_Sv -> commongrid.compute_MVBS (statistics variant)
    "cfg5api:float64": ("cfg5", ["fused_sv_mvbs_kernel<double, float, true, true, 0>"], None, 4 * 250_000 * 4096 * 12),
    "cfg2:float32": ("cfg2_f32", ["fused_sv_mvbs_kernel<float, float, true, false, 0>"], None, 4 * 500_000 * 2000 * 8),
    "cfg3:float32": ("cfg3_f32", ["sv_noise_fast_kernel", "sv_denoise_mvbs_fast_kernel", "sv_denoise_mvbs_uniform_kernel",
                                  "sv_denoise_mvbs_drift_kernel"], None, 4 * 500_000 * 2000 * 20),
    "cfg3:float64:ss2000": ("cfg3_ss2000", ["sv_noise_fast_kernel", "sv_denoise_mvbs_fast_kernel",
                                            "sv_denoise_mvbs_uniform_kernel", "sv_denoise_mvbs_drift_kernel"], None,
                            4 * 500_000 * 2000 * 32),
    "cfg4:float32": ("cfg4_f32", ["sv_complex_fft_kernel<float, float, float"], None, 2 * 200_000 * 8192 * 36),
    # the two reference calls with the Sv deferred: the statistics variant of the fused kernel inside compute_MVBS
    "api:float64": ("api", ["fused_sv_mvbs_kernel<double, float, true, true, 0>"], None, 4 * 500_000 * 2000 * 12),
    # the three reference calls of the chain: pass 1 inside remove_background_noise, pass 2 inside compute_MVBS
    "api:chain:float64": ("api_chain", ["sv_noise_fast_kernel", "sv_denoise_mvbs_fast_kernel", "sv_denoise_mvbs_uniform_kernel",
                                        "sv_denoise_mvbs_drift_kernel"], None, 4 * 500_000 * 2000 * 32),
}
# the SURVEY 8f rows (round 5): a pass launches several kernels, some of them more than once -- the traffic of a pass is
# the SUM over every dispatch of the listed kernels in the PMC run divided by the passes it made (--steps 1 --warmup 1
# --passes 2 = 4); kernels of the line's one-off setup (compute_Sv / add_depth of the resident dataset) are not listed
PASSES_IN_PMC_RUN = 4
PER_PASS = {
    "cfg2:float64:int16": ("cfg2_int16", ["fused_sv_mvbs_kernel<double, short, true"], 4 * 500_000 * 2000 * 10),
    "cfg2:float32:int16": ("cfg2_int16f32", ["fused_sv_mvbs_kernel<float, short, true"], 4 * 500_000 * 2000 * 6),
    "cfg2:float64:bins": ("cfg2_bins", ["fused_sv_mvbs_kernel<double, float, false"], 4 * 500_000 * 2000 * 4),
    "cfg2:float64:int16:bins": ("cfg2_int16bins", ["fused_sv_mvbs_kernel<double, short, false"], 4 * 500_000 * 2000 * 2),
    "cfg2:float64:sv": ("cfg2_sv", ["sv_power_piece_kernel<double", "nl_table_kernel", "d_span_kernel"], 4 * 500_000 * 2000 * 12),
    "cfg2:float32:sv": ("cfg2_sv32", ["sv_power_piece_kernel<float"], 4 * 500_000 * 2000 * 8),
    # round 6: compute_Sv -> add_depth -> compute_MVBS(range_var="depth") is ONE sweep of the raw samples (DEPTH = 1: depth
    # stays an affine description; DEPTH = 2 / next:depthw: the depth array is written beside Sv)
    "next:depth:float64": ("next_depth", ["fused_sv_mvbs_kernel<double, float, true, true, 1>"], 4 * 100_000 * 2000 * 12),
    "next:depthw:float64": ("next_depthw", ["fused_sv_mvbs_kernel<double, float, true, true, 2>"], 4 * 100_000 * 2000 * 20),
    "next:masks:float64": ("next_masks", ["range_bin_smooth", "impulse_compare", "attenuated", "pool_value", "value_",
                                          "row_interval", "row_running", "rows_check", "rows_same", "mask_and", "apply_mask",
                                          "minmax", "step_", "box_"], 4 * 100_000 * 2000 * 70),
    "next:masks2000:float64": ("next_masks2000", ["range_bin_smooth", "impulse_compare", "attenuated", "pool_value", "value_",
                                                  "row_interval", "row_running", "rows_check", "rows_same", "mask_and",
                                                  "apply_mask", "minmax", "step_", "box_", "run_"], 4 * 100_000 * 2000 * 70),
    "next:masksidx:float64": ("next_masksidx", ["range_bin_smooth", "impulse_compare", "attenuated", "pool_", "mask_and",
                                                "apply_mask", "minmax", "step_", "box_"], 4 * 100_000 * 2000 * 70),
    "next:nasc:float64": ("next_nasc", ["nasc_"], 4 * 100_000 * 2000 * 16),
}


def sum_per_kernel(d, counter):
    acc = defaultdict(float)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            if r["Counter_Name"] == counter and not r["Kernel_Name"].startswith("void at::"):
                acc[r["Kernel_Name"]] += float(r["Counter_Value"])
    if not acc:  # the raw dumps were dropped: per-kernel mean x dispatches kept in <src>/pmc_traffic.csv
        wl = os.path.basename(d).split("_", 1)[1]
        if os.path.exists(os.path.join(src, "pmc_traffic.csv")):
            for row in csv.reader(open(os.path.join(src, "pmc_traffic.csv"), newline="")):
                if row and row[0] == wl and row[2] == counter:
                    acc[row[1]] += float(row[3]) * float(row[4])
    return acc


def mean_per_kernel(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    if not acc:  # the raw dumps were dropped: the per-kernel means kept in <src>/pmc_traffic.csv (scripts/final_round.sh)
        wl = os.path.basename(d).split("_", 1)[1]
        if not os.path.exists(os.path.join(src, "pmc_traffic.csv")):
            return {}
        for row in csv.reader(open(os.path.join(src, "pmc_traffic.csv"), newline="")):
            if row and row[0] == wl and row[2] == counter:
                acc[row[1]].append(float(row[3]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
# "merge": keep the entries of the committed file that this directory does not measure again (a later line added to the
# record without repeating every pass; the entries carry the hash of the sources they were measured on)
out = json.load(open(path)) if len(sys.argv) > 2 and sys.argv[2] == "merge" else {}
for key, (wl, names, _, algo) in KERNELS.items():
    fd, wd = os.path.join(src, f"fetch_{wl}"), os.path.join(src, f"write_{wl}")
    if not (os.path.isdir(fd) and os.path.isdir(wd)):
        continue
    fe, wr = mean_per_kernel(fd, "FETCH_SIZE"), mean_per_kernel(wd, "WRITE_SIZE")
    pick = lambda k: any(n in k for n in names) and "true>" not in k.split("fft_kernel")[-1][:40]  # noqa: E731
    fkb = sum(v for k, v in fe.items() if pick(k))
    wkb = sum(v for k, v in wr.items() if pick(k))
    if fkb == 0 and wkb == 0:
        continue
    e = {"bytes_per_launch": 2 * fkb * 1024 + wkb * 1024, "fetch_size_kb_raw": fkb, "write_size_kb_raw": wkb,
         "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE x1"
                       + ("; the kernels of a pass summed" if len(names) > 1 else ""),
         "algorithmic_bytes": algo, "csrc_sha16": csrc_hash(),
         "source": f"profiles/r06_pmc_traffic.csv (bench.py --workload {wl.replace('_', ':')})"}
    k0f = [v for k, v in fe.items() if "power_coef_ek_kernel" in k]
    k0w = [v for k, v in wr.items() if "power_coef_ek_kernel" in k]
    if wl in ("cfg2", "cfg3", "cfg2_f32", "cfg3_f32", "cfg3_ss2000") and k0f and k0w:  # K0 reads 5 x (C, P) f64 + small tables, writes 64 B per (c, p)
        cp = 4 * 500_000
        e["calibration"] = {"kernel": "power_coef_ek_kernel (known 80 MB read, 128 MB write)",
                            "FETCH_SIZE_ratio_raw": k0f[0] * 1024 / (cp * 40.0), "WRITE_SIZE_ratio_raw": k0w[0] * 1024 / (cp * 64.0)}
    out[key] = e
    print(key, "traffic / algorithmic = %.4f" % (e["bytes_per_launch"] / algo), e.get("calibration"))
for key, (wl, names, algo) in PER_PASS.items():
    fd, wd = os.path.join(src, f"fetch_{wl}"), os.path.join(src, f"write_{wl}")
    if not (os.path.isdir(fd) and os.path.isdir(wd)):
        continue
    fe, wr = sum_per_kernel(fd, "FETCH_SIZE"), sum_per_kernel(wd, "WRITE_SIZE")
    fkb = sum(v for k, v in fe.items() if any(n in k for n in names)) / PASSES_IN_PMC_RUN
    wkb = sum(v for k, v in wr.items() if any(n in k for n in names)) / PASSES_IN_PMC_RUN
    if fkb == 0 and wkb == 0:
        continue
    out[key] = {"bytes_per_launch": 2 * fkb * 1024 + wkb * 1024, "fetch_size_kb_raw": fkb, "write_size_kb_raw": wkb,
                "correction": "FETCH_SIZE x2, WRITE_SIZE x1; every dispatch of the pass's kernels summed, per pass",
                "algorithmic_bytes": algo, "csrc_sha16": csrc_hash(),
                "source": f"profiles/r06_pmc_traffic.csv (bench.py --workload {wl.replace('_', ':')})"}
    print(key, "traffic / algorithmic = %.4f" % (out[key]["bytes_per_launch"] / algo))
json.dump(out, open(path, "w"), indent=1)
print("wrote", path)
