"""Condense the output of tools/collect_profiles.sh: kernel-stats CSVs are copied as they are, the --pmc passes
become one JSON (last launch of each finrom kernel).  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB;
hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts half of the bytes of a
coalesced streaming read, /opt/skills/guides/MI355X_MICROARCH.md).
usage: python tools/pmc_summary.py gpurun_out/prof_<tag> [workload_key]"""
import csv
import glob
import json
import os
import re
import sys

out = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "five/m12/r80/S100000"
raw = {}
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            m = re.search(r"finrom::(?:\(anonymous namespace\)::)?(\w+)", row["Kernel_Name"])
            if not m:
                continue
            raw.setdefault(m.group(1), {})[row["Counter_Name"]] = float(row["Counter_Value"])     # last launch wins
summary = {"workload_key": key,
           "note": "rocprofv3 --pmc passes (one counter group per run, no tracing, halves serialised with FINROM_NO_OVERLAP=1) of "
                   "`bench.py --steps 1 --warmup 1 --cpu-samples 0 --no-profile`; values of the last launch of each kernel. "
                   "FETCH_SIZE/WRITE_SIZE in KiB; hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE doubled "
                   "as MI355X_MICROARCH.md prescribes for gfx950; WRITE_SIZE exact).",
           "raw": raw, "hbm_bytes_per_launch": {}, "l2_hit_rate": {}, "mfma_busy_frac": {}}
slot_of = {"fom_band_kernel": "fom_chol_solve", "fom_vm_kernel": "fom_chol_solve", "fom_bwd_kernel": "fom_chol_solve", "rom_gram_kernel": "rom_proj_mfma", "rom_gram_store_kernel": "rom_proj_mfma", "fom_assemble_kernel": "fom_assemble", "rom_proj_kernel": "rom_proj_mfma",
           "rom_proj_lds_kernel": "rom_proj_mfma", "rom_proj_kernel_r80": "rom_proj_mfma", "rom_proj_single_kernel": "rom_proj_mfma", "rom_solve_kernel": "rom_reduced_solve", "subfin_avg_kernel": "subfin_avg",
           "pack_kernel": "pack"}
for k, c in raw.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:      # kernels that share a timer slot (FOM forward + backward) add up
        summary["hbm_bytes_per_launch"][slot_of.get(k, k)] = summary["hbm_bytes_per_launch"].get(slot_of.get(k, k), 0) + \
            int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
        summary.setdefault("hbm_bytes_per_kernel", {})[k] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
    if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
        summary["l2_hit_rate"][k] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    if c.get("GRBM_GUI_ACTIVE", 0) > 0 and c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
        summary["mfma_busy_frac"][k] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (c["GRBM_GUI_ACTIVE"] / 8), 4)
with open(os.path.join(out, "pmc_summary.json"), "w") as fh:
    json.dump(summary, fh, indent=1)
print(json.dumps({k: summary[k] for k in ("hbm_bytes_per_launch", "l2_hit_rate", "mfma_busy_frac")}, indent=1))
for mode in ("overlapped", "serial"):
    for f in glob.glob(os.path.join(out, mode, "**", "*kernel_stats.csv"), recursive=True):
        print("==", mode, f)
        with open(f) as fh:
            for i, line in enumerate(fh):
                if i < 8:
                    print(line.rstrip()[:220])
