"""Condense the output of tools/collect_profiles.sh: one JSON per workload (last launch of each finrom kernel) from the --pmc
passes, plus the check whether rocprofv3 lets the two halves overlap while it collects counters.  FETCH_SIZE / WRITE_SIZE are
reported by rocprofv3 in KiB; hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts half of
the bytes of a coalesced streaming read, /opt/skills/guides/MI355X_MICROARCH.md; WRITE_SIZE exact).
usage: python tools/pmc_summary.py gpurun_out/prof_<tag>"""
import csv
import glob
import json
import os
import re
import sys

out = sys.argv[1]
WORKLOADS = {"headline": "five/m12/r80/S100000", "c3": "nine/m12/r120/S100000", "c4": "field/m20/r200/S20000",
             "c4full": "field/m20/r200/S125000"}
SLOT_OF = {"fom_band_kernel": "fom_chol_solve", "fom_band_ldsw_kernel": "fom_chol_solve", "fom_vm_kernel": "fom_chol_solve",
           "fom_bwd_kernel": "fom_chol_solve", "fom_small_kernel": "fom_chol_solve", "rom_gram_kernel": "rom_proj_mfma",
           "rom_gram_store_kernel": "rom_proj_mfma", "fom_assemble_kernel": "fom_assemble", "rom_proj_kernel": "rom_proj_mfma",
           "rom_proj_lds_kernel": "rom_proj_mfma", "rom_proj_single_kernel": "rom_proj_mfma", "rom_solve_kernel": "rom_reduced_solve",
           "rom_chol_blocked_kernel": "rom_reduced_solve", "rom_subst_blocked_kernel": "rom_reduced_solve",
           "subfin_avg_kernel": "subfin_avg", "subfin_avg_small_kernel": "subfin_avg", "pack_kernel": "pack", "sampler_kernel": "sampler_gemm_exp", "sampler_gemm_kernel": "sampler_gemm_exp",
           "sampler_small_kernel": "sampler_gemm_exp"}


def kname(row):
    m = re.search(r"finrom::(?:\(anonymous namespace\)::)?(\w+)", row["Kernel_Name"])
    return m.group(1) if m else None


for sub, key in WORKLOADS.items():
    raw = {}
    for f in glob.glob(os.path.join(out, sub, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = kname(row)
                if k:
                    raw.setdefault(k, {})[row["Counter_Name"]] = float(row["Counter_Value"])     # last launch wins
    if not raw:
        continue
    summary = {"workload_key": key, "commit": os.environ.get("FINROM_COMMIT"),
               "note": "rocprofv3 --pmc passes (one counter group per run, no tracing, halves in turn: FINROM_NO_OVERLAP=1) of "
                       "`bench.py --steps 1 --warmup 1 --cpu-samples 0 --no-profile` + the workload's flags; values of the last launch "
                       "of each kernel.  FETCH_SIZE/WRITE_SIZE in KiB; hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                       "(FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; WRITE_SIZE exact).",
               "raw": raw, "hbm_bytes_per_launch": {}, "hbm_bytes_per_kernel": {}, "l2_hit_rate": {}, "mfma_busy_frac": {}}
    for k, c in raw.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:      # kernels that share a timer slot add up
            b = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
            slot = SLOT_OF.get(k, k)
            summary["hbm_bytes_per_launch"][slot] = summary["hbm_bytes_per_launch"].get(slot, 0) + b
            summary["hbm_bytes_per_kernel"][k] = b
        if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
            summary["l2_hit_rate"][k] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        if c.get("GRBM_GUI_ACTIVE", 0) > 0 and c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
            summary["mfma_busy_frac"][k] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (c["GRBM_GUI_ACTIVE"] / 8), 4)
    with open(os.path.join(out, f"pmc_summary_{sub}.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    print("==", sub, key)
    print(json.dumps({k: summary[k] for k in ("hbm_bytes_per_launch", "l2_hit_rate", "mfma_busy_frac")}, indent=1))

# does the profiler let the two halves of finrom_solve_pairs overlap while it collects counters?
trace = glob.glob(os.path.join(out, "pmc_overlap_check", "**", "*kernel_trace.csv"), recursive=True)
check = {"question": "with --pmc, do the projection and the band sweep of one finrom_solve_pairs call run side by side?"}
if trace:
    rows = []
    with open(trace[0]) as fh:
        for row in csv.DictReader(fh):
            k = kname(row)
            if k in ("rom_proj_single_kernel", "fom_band_kernel"):
                rows.append((k, int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
    proj = [r for r in rows if r[0] == "rom_proj_single_kernel"][-1:]
    band = [r for r in rows if r[0] == "fom_band_kernel"][-1:]
    if proj and band:
        (_, p0, p1), (_, b0, b1) = proj[0], band[0]
        ov = max(0, min(p1, b1) - max(p0, b0))
        check.update({"projection_ms": (p1 - p0) / 1e6, "sweep_ms": (b1 - b0) / 1e6, "overlap_ms": ov / 1e6,
                      "answer": "serialised by the profiler: counters of the side-by-side configuration cannot be collected with "
                                "rocprofv3 on this stack" if ov < 0.05 * (b1 - b0) else "they overlap: the counters below are of the side-by-side run"})
    raw = {}
    for f in glob.glob(os.path.join(out, "pmc_overlap_check", "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = kname(row)
                if k in ("rom_proj_single_kernel", "fom_band_kernel"):
                    raw.setdefault(k, {})[row["Counter_Name"]] = float(row["Counter_Value"])
    check["counters_last_launch"] = raw
else:
    check["answer"] = "the pass with --kernel-trace and --pmc together produced no trace (see pmc_overlap_check.log)"
with open(os.path.join(out, "pmc_overlap_check.json"), "w") as fh:
    json.dump(check, fh, indent=1)
print("== overlap check:", check.get("answer"), {k: check.get(k) for k in ("projection_ms", "sweep_ms", "overlap_ms")})
for mode in ("overlapped", "serial", "c3", "c4"):
    for f in glob.glob(os.path.join(out, mode, "**", "*kernel_stats.csv"), recursive=True):
        print("==", mode, f)
        with open(f) as fh:
            for i, line in enumerate(fh):
                if i < 7:
                    print(line.rstrip()[:200])
