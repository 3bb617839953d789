"""bench.py: the N = 1 line's contract and the N > 1 path (torch.distributed.run launch, barrier + max-over-ranks timing, rank-0 JSON line,
bucketed gradient all-reduce on the full-size model) exercised with two ranks sharing the test box's one GPU
over gloo.  The numbers are meaningless; the code path is the one the driver runs on 2 / 4 / 8 GPUs."""

import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run_bench(cmd, env, attempts=2, timeout=300):
    """Two ranks time-sharing ONE GPU over gloo is a test-only arrangement (normally 15-20 s).  Once in ~10 full-suite runs of
    round 4 such a launch sat until its 900-s limit on a fresh box and could not be reproduced in 8 repetitions of the same
    command (alone, and behind the same preceding test files): a stuck launch is cut after `timeout` seconds and tried once
    more; if it sticks twice the test fails with both outputs."""
    import signal
    from types import SimpleNamespace

    notes = []
    for k in range(attempts):
        # own session: on a cut, the launcher AND its rank processes go (they would otherwise keep the GPU)
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT,
                                start_new_session=True)
        try:
            out, err = proc.communicate(timeout=timeout)
            return SimpleNamespace(returncode=proc.returncode, stdout=out, stderr=err)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            out, err = proc.communicate()
            notes.append(f"attempt {k}: no result after {timeout} s\nstdout: {(out or '')[-1500:]}\nstderr: {(err or '')[-1500:]}")
    pytest.fail("bench.py --gpus 2 did not finish:\n" + "\n".join(notes))


def test_bench_two_ranks_one_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, TDX_BENCH_BACKEND="gloo", TDX_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "1",
           "--no-cpu-baseline", "--sample-steps", "3", "--sample-batch", "1"]
    out = _run_bench(cmd, env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 2
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["config"]["parallelism"] == "dp2"
    assert abs(d["value"] - 2 * 1 * 192 * 64 * 48 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6  # whole-job aggregate
    # overlap report of the gradient all-reduce (step vs step without the collective vs the collective alone)
    ov = d["extra"]["overlap"]
    assert ov["ms_step"] > 0 and ov["ms_step_without_allreduce"] > 0 and ov["ms_allreduce_alone"] > 0
    assert abs(ov["payload_MB"] - 55246788 * 4 / 1e6) < 1.0 and len(ov["buckets"]) >= 4
    # BASELINE configs[3]: trajectories sharded over the ranks, aggregate samples/s
    sm = d["extra"]["sampling"]
    assert sm["trajectories"] == 2 and sm["trajectories_per_gpu"] == 1 and sm["ddpm_samples_per_s_T1000"] > 0


def test_bench_bare_command_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (the command the driver uses for N = 1, with N = 2): bench.py
    starts torch.distributed.run itself as a child process before it touches the GPU, relays rank 0's JSON line and the
    exit code; the overlap report carries the HIP-event figures (backward span, exposed all-reduce wait)."""
    env = dict(os.environ, TDX_BENCH_BACKEND="gloo", TDX_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "1",
           "--no-cpu-baseline", "--sample-steps", "0"]
    out = _run_bench(cmd, env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["value"] > 0
    ov = d["extra"]["overlap"]
    assert ov["ms_backward"] > 0 and ov["ms_exposed"] >= 0 and "HIP events" in ov["method"]
    # the per-bucket report is populated, and the buckets were enqueued in backward order
    assert ov["hidden_fraction"] is not None and len(ov["per_bucket"]) == len(ov["buckets"]) >= 4
    at = [b["enqueued_at_ms"] for b in ov["per_bucket"]]
    assert all(a is not None and a > 0 for a in at) and at == sorted(at), at
    assert all(b["alone_ms"] > 0 and b["exposed_ms"] >= 0 and b["hidden_fraction"] is not None for b in ov["per_bucket"])
    # item 7b: every rank reports its host enqueue time per step and where its threads are pinned
    assert [r["rank"] for r in ov["per_rank"]] == [0, 1] and all(r["host_enqueue_ms_per_step"] > 0 for r in ov["per_rank"])
    aff = [r["host_affinity"] for r in ov["per_rank"]]
    assert all("pinned" in a for a in aff)
    if all(a["pinned"] for a in aff):
        assert aff[0]["cores"] != aff[1]["cores"] or aff[0]["numa_node"] != aff[1]["numa_node"]
    # item 7a: the captured step ran under data parallelism in the same call
    cs = ov["captured_step"]
    assert "error" not in cs and cs["ms_per_step"] > 0, cs


def test_bench_data_parallel_legs_over_rccl_at_world_size_one():
    """TDX_BENCH_FORCE_DDP=1: the whole N > 1 path of bench.py on the REAL backend (RCCL communicator, gradient hooks with the
    asynchronous all-reduce on RCCL's stream, 224-CU persistent kernels, overlap report, sharded sampling leg captured right
    behind an RCCL barrier, the captured data-parallel step with its host-polled collectives) with the one GPU this box has.
    The two-rank tests above use gloo; what only RCCL does (its watchdog thread, its own stream, its kernels beside the
    persistent conv kernels) is covered here."""
    env = dict(os.environ, TDX_BENCH_FORCE_DDP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "TDX_BENCH_BACKEND", "TDX_BENCH_ONE_DEVICE"):
        env.pop(k, None)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        env["MASTER_PORT"] = str(s.getsockname()[1])
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
           "--sample-steps", "3", "--sample-batch", "1"]
    out = _run_bench(cmd, env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["global_batch"] == 6
    # the kernels run at their usual speed beside the communicator (0.39-0.42 of the MFMA peak on the forward convs): a
    # high-priority side stream, free on one GPU, took them to 0.30 and the step from 21.2 to 33.0 ms as soon as an RCCL
    # communicator was alive -- only this dry run showed it
    # (0.348 with TDX_DETERMINISTIC=1 exported: the statistics pass then sits inside the forward conv's event bracket)
    assert d["roofline"]["frac"] > 0.325, d["roofline"]["frac"]
    ov = d["extra"]["overlap"]
    assert ov["persistent_cus"] == 224 and ov["nccl_max_nchannels"] == "32", ov
    assert ov["ms_allreduce_alone"] > 0 and ov["ms_backward"] > 0 and len(ov["per_bucket"]) == len(ov["buckets"]) >= 4
    at = [b["enqueued_at_ms"] for b in ov["per_bucket"]]
    assert all(a is not None and a > 0 for a in at) and at == sorted(at), at
    cs = ov["captured_step"]
    assert "error" not in cs and cs["ms_per_step"] > 0, cs
    assert d["extra"]["sampling"]["ddpm_samples_per_s_T1000"] > 0


def test_bench_single_gpu_line_keeps_its_contract():
    """`python bench.py` at N = 1 (short: no CPU baseline, no extra legs, traffic from the committed PMC file): ONE JSON line with
    the keys and meanings the driver reads -- BASELINE configs[1]'s metric on its workload, whole-job voxels/s consistent with
    ms_per_step, the roofline block of the dominant kernel measured in the timed region."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TDX_BENCH_FORCE_DDP", "TDX_BENCH_BACKEND",
              "TDX_BENCH_ONE_DEVICE"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-extra",
           "--no-live-traffic"]
    out = _run_bench(cmd, env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    d = json.loads(lines[0])
    baseline = json.loads((ROOT / "BASELINE.json").read_text())  # "U-Net fwd+bwd voxels/sec + DDPM samples/sec, 192x64x48x4, ..."
    assert d["metric"].startswith("U-Net fwd+bwd voxels/sec") and baseline["metric"].startswith("U-Net fwd+bwd voxels/sec")
    assert "192x64x48x4" in d["metric"] and d["unit"] == "voxels/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and "configs[1]" in d["config"]["workload"]
    assert d["config"]["global_batch"] == 6 and d["config"]["grid"] == [192, 64, 48] and d["config"]["parallelism"] == "dp1"
    assert abs(d["value"] - 6 * 192 * 64 * 48 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.2 < r["frac"] < 1.0
    assert r["launches"] == 16 * 4 and r["avg_launch_ms"] > 0  # the brick / ring forward launches of the 4 timed steps
    assert r["algorithmic_bytes_per_launch"] > 1e8 and (r["traffic"] is None or r["traffic"] > r["algorithmic_bytes_per_launch"] * 0.9)
    assert 5.0 < d["ms_per_step"] < 200.0 and abs(d["loss"]) < 10.0
