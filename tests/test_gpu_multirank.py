"""The N > 1 path with REAL product output on the one-GPU box: `python bench.py --gpus 2 --backend gloo` starts two rank
processes that share cuda:0, each solves its shard of the globally keyed sample stream through the HIP library, the QoI pairs
are gathered over gloo, and the gathered array must equal -- bit for bit -- what one process computes for the same 2 x S
samples (results independent of the GPU count).  On an 8-GPU node the same entry runs with --backend nccl."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(argv):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("params,extra", [("five", []), ("field", ["--m", "4", "--r", "8"])])
def test_two_ranks_reproduce_the_single_process_outputs(params, extra):
    common = ["--steps", "1", "--warmup", "1", "--cpu-samples", "0", "--no-profile", "--params", params] + extra
    two = _bench(["--gpus", "2", "--backend", "gloo", "--samples", "3000"] + common)
    one = _bench(["--gpus", "1", "--samples", "6000"] + common)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["failed_samples"] == 0 and one["config"]["failed_samples"] == 0
    assert two["gathered_sha256"] == one["gathered_sha256"]
    assert two["value"] > 0 and two["scaling"] == "weak"
