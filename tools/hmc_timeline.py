"""Kernel durations of the HMC leapfrog step from a rocprofv3 kernel trace of `bench.py --workload hmc` (graph replay):
    cd /tmp; rocprofv3 --kernel-trace --output-format csv -d OUT -o run -- python3 bench.py --workload hmc --steps 400 --warmup 50 --cpu-samples 0
    python tools/hmc_timeline.py OUT/run_kernel_trace.csv
Prints the median duration of every kernel that runs once per step (library kernels by name, torch's elementwise kernels summed)
and a window of consecutive dispatches with the gaps between them (under the profiler dispatches are serialised: the gaps are the
profiler's, the durations are the kernels')."""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    return re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", n))[:70]


dur = defaultdict(list)
for r in rows:
    dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
steps = max(len(v) for k, v in dur.items() if "finrom::mlp_backward" in k)
lib = tor = 0.0
print(f"{steps} leapfrog steps in the trace")
for k, v in sorted(dur.items(), key=lambda kv: -len(kv[1])):
    v = sorted(v)
    per_step = len(v) / steps
    if per_step < 0.09:                                  # (0.1 per step: the proposal's begin / end kernels)
        continue
    med = v[len(v) // 2]
    print(f"{per_step:5.2f} per step, median {med:7.2f} us  {k}")
    if "finrom::" in k:
        lib += med * round(per_step)
    else:
        tor += med * per_step
print(f"per step: library kernels {lib:.1f} us, torch elementwise {tor:.1f} us (serialised by the profiler)")
mid = len(rows) // 2
for i in range(mid, mid + 16):
    r, nx = rows[i], rows[i + 1]
    print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.2f} us, then {(int(nx['Start_Timestamp']) - int(r['End_Timestamp'])) / 1e3:6.2f}  {short(r['Kernel_Name'])}")
