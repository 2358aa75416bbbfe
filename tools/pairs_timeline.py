"""Start / end of every kernel of the last finrom_solve_pairs calls in a rocprofv3 kernel trace of bench.py (pairs workload):
    cd /tmp; rocprofv3 --kernel-trace --output-format csv -d OUT -o run -- python3 bench.py --steps 3 --warmup 1 --cpu-samples 0 --no-host-io --no-other
    python tools/pairs_timeline.py OUT/run_kernel_trace.csv
Times in ms relative to the start of the second-to-last projection kernel: what sits between one step's projection and the next's."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    return re.sub(r"\(.*", "", re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", n)))[:60]


proj = [i for i, r in enumerate(rows) if "rom_proj" in r["Kernel_Name"]]
i0 = proj[-2]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[max(0, i0 - 8):]:
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print(f"{a:9.3f} .. {b:9.3f}  ({b - a:7.3f} ms)  {short(r['Kernel_Name'])}")
