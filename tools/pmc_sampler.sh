#!/bin/bash
# usage (on the GPU box): bash tools/pmc_sampler.sh <tag> [S] [n]
# rocprofv3 --pmc passes (one counter group per run) over tools/sampler_bench.py for each variant of the sampler GEMM; prints per
# kernel: HBM bytes per launch = (2 FETCH_SIZE + WRITE_SIZE) KiB (gfx950: MI355X_MICROARCH.md), L2 hit rate, MFMA pipe busy fraction
tag=$1; S=${2:-20000}; n=${3:-4101}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
out=gpurun_out/pmc_sampler_$tag
mkdir -p $out
for v in small_64x64 gemm_256x128 pad_top2 pad_top3 gemm_256x128_nopad; do
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --output-format csv -d $out/$v/g$i -o run -- python tools/sampler_bench.py $S $n $v 1 > $out/$v.g$i.log 2>&1 || { echo "$v group $i failed"; tail -3 $out/$v.g$i.log; }
  done
done
python - "$out" <<'PY'
import csv, glob, sys, re, os, json
res = {}
for v in ("small_64x64", "gemm_256x128", "pad_top2", "pad_top3", "gemm_256x128_nopad"):
    raw = {}
    for f in glob.glob(os.path.join(sys.argv[1], v, "g*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"finrom::(?:\(anonymous namespace\)::)?(sampler\w+)", row["Kernel_Name"])
            if m: raw.setdefault(m.group(1), {})[row["Counter_Name"]] = float(row["Counter_Value"])     # last launch wins
    for k, c in raw.items():
        r = {"raw": c}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c: r["hbm_bytes_per_launch"] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
        if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0: r["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        if c.get("GRBM_GUI_ACTIVE", 0) > 0: r["mfma_busy_frac"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (c["GRBM_GUI_ACTIVE"] / 8), 4)
        res[v + ":" + k] = r
json.dump(res, open(os.path.join(sys.argv[1], "summary.json"), "w"), indent=1)
for k, r in res.items():
    print(k, {a: b for a, b in r.items() if a != "raw"})
PY
