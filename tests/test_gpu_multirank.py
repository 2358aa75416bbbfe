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


@pytest.mark.parametrize("mode", ["device", "device-torch"])
def test_hmc_chains_sharded_over_two_ranks_walk_the_single_process_paths(mode):
    """configs[4] ("4 chains ... sharded one chain/GPU") with two ranks sharing cuda:0 over gloo: rank g owns chains g, g + 2; every
    chain must end where it ends in the one-process run (4 chains in lockstep) -- a chain's sums do not depend on which chains share
    its launches -- with the same accept counts and the same per-proposal trace: the gathered checksums are equal."""
    common = ["--workload", "hmc", "--chains", "4", "--steps", "60", "--warmup", "10", "--cpu-samples", "0", "--no-profile",
              "--hmc-trace", "--hmc-mode", mode, "--hmc-eps", "0.03"]
    two = _bench(["--gpus", "2", "--backend", "gloo"] + common)
    one = _bench(["--gpus", "1"] + common)
    assert two["n_gpus"] == 2 and two["chains_gathered"] == one["chains_gathered"] == 4
    assert two["config"]["fused_leapfrog"] == (mode == "device") and two["config"]["hip_graph"] is True
    assert sum(one["config"]["accepted_all_chains"]) > 0
    assert two["config"]["accepted_all_chains"] == one["config"]["accepted_all_chains"]
    assert two["chains_sha256"] == one["chains_sha256"] and two["trace_sha256"] == one["trace_sha256"] is not None


def test_one_rank_rccl_process_group_and_device_gather():
    """The `nccl` backend (= RCCL on ROCm) executed for real before the first multi-GPU run: one rank started by
    torch.distributed.run (launched before anything touches the GPU), --force-pg creates the RCCL communicator with
    device_id = cuda:0 and runs the per-step all_gather_into_tensor + barrier + all_reduce on DEVICE tensors at world = 1; the
    gathered checksum must equal the plain single-process run's."""
    common = ["--steps", "2", "--warmup", "1", "--cpu-samples", "0", "--no-profile", "--samples", "3000"]
    plain = _bench(["--gpus", "1"] + common)
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = ["--gpus", "1", "--backend", "nccl", "--force-pg"] + common
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py")], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", BENCH_ARGV=json.dumps(argv)))
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    forced = json.loads(lines[0])
    assert forced["process_group"] == {"backend": "nccl", "world": 1, "forced": True}
    assert forced["gathered_sha256"] == plain["gathered_sha256"]
    assert forced["config"]["failed_samples"] == 0 and forced["value"] > 0
    # and without the launcher: bench.py --force-pg makes its own one-rank rendezvous
    alone = _bench(argv)
    assert alone["process_group"]["backend"] == "nccl" and alone["gathered_sha256"] == plain["gathered_sha256"]
    # the default lets the gather of step i run beside the kernels of step i + 1 (joined before the clock stops); --sync-gather
    # makes every step wait for its own: the same rows either way
    assert _bench(argv + ["--sync-gather"])["gathered_sha256"] == plain["gathered_sha256"]


def test_c_abi_gather_one_rank():
    """finrom_comm_* (the gather at the end for callers without torch): RCCL loaded by the library with dlopen, a one-rank
    communicator on cuda:0, an all-gather between device buffers allocated through the C ABI.  (Two ranks cannot share the one
    GPU of this box -- RCCL rejects duplicate devices -- so the N > 1 form of this route is exercised by construction only:
    it is the same ncclAllGather call with nranks > 1.)"""
    import numpy as np
    from bayesianinferencedl_amd import _ffi
    from bayesianinferencedl_amd.distributed import FinromComm
    _ffi.check(_ffi.lib().finrom_set_device(0))
    uid = FinromComm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = FinromComm(0, 1, uid)
    x = np.random.default_rng(0).standard_normal((1000, 18))
    src = _ffi.DeviceBuffer.from_numpy(x)
    dst = _ffi.DeviceBuffer(x.nbytes); dst.zero()
    comm.gather(src.ptr, x.size, dst.ptr)
    _ffi.check(_ffi.lib().finrom_stream_sync(None))
    assert np.array_equal(dst.to_numpy(x.shape), x)
    comm.close()
    h = __import__("ctypes").c_void_p()
    assert _ffi.lib().finrom_comm_init(__import__("ctypes").byref(h), 2, 2, uid) == -1      # rank out of range
