"""What does a node of a replayed HIP graph cost when it is one of this library's kernels, and when it is one of torch's?
Captures chains of N launches -- finrom_sub only / torch.sub only / alternating -- and times the replay with events.
(The HMC leapfrog step's timeline, tools/hmc_timeline.py, shows ~10 us between two library kernels and none between torch's.)
    python tools/graph_gap_probe.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesianinferencedl_amd import _ffi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lib = _ffi.lib()
_ffi.check(lib.finrom_set_device(0))
COUNT = int(sys.argv[2]) if len(sys.argv) > 2 else 4 * 1597
a = torch.randn(COUNT, dtype=torch.float64, device="cuda")
b = torch.randn_like(a)
out = torch.empty_like(a)


def ours():
    _ffi.check(lib.finrom_sub(a.data_ptr(), b.data_ptr(), a.numel(), out.data_ptr(), torch.cuda.current_stream().cuda_stream))


def theirs():
    torch.sub(a, b, out=out)


def chain(fs):
    def run():
        for i in range(N):
            fs[i % len(fs)]()
    return run


for name, fs in (("finrom_sub only", [ours]), ("torch.sub only", [theirs]), ("alternating", [ours, theirs])):
    run = chain(fs)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    tg = e0.elapsed_time(e1) / 20 / N * 1e3
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    te = e0.elapsed_time(e1) / 20 / N * 1e3
    print(f"count {COUNT:9d} {name:18s}: {tg:6.2f} us per node in a replayed graph, {te:6.2f} us per launch in stream order", flush=True)
