#!/bin/bash
# usage (on the GPU box): bash tools/pmc_probe.sh <tag> <what: rom|fom> "<counters>" ["<counters>" ...]
# one rocprofv3 --pmc pass per counter group over tools/proj_bench.py; prints the per-kernel values of the last launch
tag=$1; what=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
out=gpurun_out/pmc_$tag
mkdir -p $out
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/g$i -o run -- python tools/proj_bench.py 80 100000 1 $what > $out/g$i.log 2>&1 || { echo "group $i failed"; tail -3 $out/g$i.log; }
done
python - "$out" <<'PY'
import csv, glob, sys, re, os
raw = {}
for f in glob.glob(os.path.join(sys.argv[1], "g*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"finrom::(?:\(anonymous namespace\)::)?(\w+)", row["Kernel_Name"])
        if m: raw.setdefault(m.group(1), {})[row["Counter_Name"]] = float(row["Counter_Value"])
for k, c in raw.items():
    print(k, {n: (int(v) if v == int(v) else v) for n, v in sorted(c.items())})
PY
