"""bench.py prints ONE JSON line with the contract's keys (small workload so it runs in seconds)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--samples", "4096", "--steps", "2", "--warmup", "1",
                        "--cpu-samples", "8"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "pairs/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 4096 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and 0 < rf["frac"] <= 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and "traffic" in rf
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == "pairs/s"
    assert d["config"]["failed_samples"] == 0
    _flops_below_peak(d)
    assert rf["traffic"] is None or (rf["traffic_source"] and rf["traffic_source"]["file"].startswith("profiles/"))


def _flops_below_peak(d):
    """config.flops_per_pair (SURVEY 8(d)'s formula for the form actually run) x samples / step time can never exceed the fp64
    matrix peak (the field line once claimed 118 TFLOP/s because the affine-assembly term was charged per NODE)."""
    S = d["config"]["samples_per_gpu"]
    tflops = d["config"]["flops_per_pair"] * S / (d["ms_per_step"] * 1e-3) / 1e12
    assert 0 < tflops <= 78.6, tflops


@pytest.mark.parametrize("argv", [["--params", "nine", "--r", "120", "--samples", "4096"],
                                  ["--params", "field", "--m", "20", "--r", "200", "--samples", "1024"],
                                  ["--projection", "offline_online", "--samples", "8192"],
                                  ["--params", "nine", "--r", "120", "--samples", "4096", "--projection", "offline_online"]])
def test_bench_roofline_is_a_fraction_for_every_config(argv):
    """The roofline object is chosen by kernel (projection / blocked reduced solve / FOM sweep ...): whatever kernel dominates a
    configuration, 0 < frac <= 1 (the interpreter's no-cache byte model applied to another kernel gave 1.21)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-samples", "0",
                        "--no-other", "--no-host-io"] + argv, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    rf = d["roofline"]
    assert rf is not None and 0 < rf["frac"] <= 1.0, rf
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["kernel"] in d["kernels_avg_ms"] and "model" in rf
    assert d["config"]["failed_samples"] == 0
    _flops_below_peak(d)


def test_cu_masked_fom_stream_gives_the_same_outputs():
    """Beside the one-wave projection kernel the FOM sweep of a large batch runs on a library stream masked to three CUs per
    shader engine (finrom_solve_pairs).  Where the kernels run must not change what they compute: the gathered QoI pairs
    of the masked default, of an unmasked run (FINROM_FOM_CUS=0) and of another mask (64 CUs) have the same checksum."""
    def run(env, *extra):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--samples", "20000", "--steps", "2", "--warmup", "1",
                            "--cpu-samples", "0", "--no-other", "--no-host-io", "--no-profile", *extra], capture_output=True, text=True,
                           timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout + r.stderr
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    masked, plain, other = run({}), run({"FINROM_FOM_CUS": "0"}), run({"FINROM_FOM_CUS": "64"})
    assert masked["config"]["failed_samples"] == 0 and masked["config"]["stream"] == "own"
    assert masked["gathered_sha256"] == plain["gathered_sha256"] == other["gathered_sha256"]
    # ... nor which stream carries which half: the steps on torch's null stream (the masked stream is a blocking one: the ROM half
    # then runs on the library's side stream), the ROM half forced to the side stream, the whole FOM half on the masked stream
    for env, extra in (({}, ("--stream", "default")), ({"FINROM_ROM_ON_SIDE": "1"}, ()), ({"FINROM_FOM_PREPASS_MASKED": "1"}, ())):
        assert run(env, *extra)["gathered_sha256"] == masked["gathered_sha256"], (env, extra)
