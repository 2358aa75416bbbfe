"""Summarise a FINROM_TRACE dump: per-kernel workgroup start/end times and per-CU residency over time.
usage: python tools/trace_report.py <prefix>"""
import sys
import numpy as np

prefix = sys.argv[1]
tr = {}
for kind in ("fom", "proj"):
    try:
        a = np.fromfile(f"{prefix}.{kind}.bin", dtype=np.int64).reshape(-1, 6)
    except FileNotFoundError:
        continue
    a = a[(a[:, 0] != 0) & (a[:, 1] != 0)]
    if len(a):
        tr[kind] = a
t0 = min(a[:, 0].min() for a in tr.values())
for kind, a in tr.items():
    st, en = (a[:, 0] - t0) * 1e-5, (a[:, 1] - t0) * 1e-5          # ms
    hw, xcc = a[:, 2], a[:, 3] & 15
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
    dur = en - st
    print(f"{kind}: {len(a)} workgroups, first start {st.min():.3f} ms, last start {st.max():.3f}, last end {en.max():.3f}; "
          f"duration min/med/max {dur.min():.3f}/{np.median(dur):.3f}/{dur.max():.3f} ms; distinct CUs {len(np.unique(cu))}")
    edges = np.linspace(0, max(x[:, 1].max() - t0 for x in tr.values()) * 1e-5, 21)
    res = [((st <= t) & (en > t)).sum() for t in edges]
    print("   resident workgroups at", " ".join(f"{t:.0f}ms:{r}" for t, r in zip(edges, res)))
    percu = np.bincount(np.unique(cu, return_inverse=True)[1][(st <= edges[3]) & (en > edges[3])])
    print("   per-CU resident at t=%.1f ms: min %d max %d mean %.2f" % (edges[3], percu.min(), percu.max(), percu.mean()))
    mhz = (a[:, 5] - a[:, 4]) / np.maximum(a[:, 1] - a[:, 0], 1) * 100.0       # shader ticks per 10 ns tick
    print("   shader clock over the workgroups: min %.0f median %.0f max %.0f MHz; median cycles per workgroup %.0f"
          % (mhz.min(), np.median(mhz), mhz.max(), np.median(a[:, 5] - a[:, 4])))
    sim = (hw >> 4) & 3
    print("   SIMD histogram of wave 0:", np.bincount(sim, minlength=4))
