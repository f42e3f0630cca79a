"""CPU: `python bench.py --gpus 2` from a bare shell launches its own ranks (bench.py:launch_ranks) and runs the
multi-rank step loop end to end — process-group bring-up, barrier-bracketed timing, the DetectionGather slots, the
overflow accumulator, ONE JSON line from rank 0 — on gloo with the detector stubbed (S2A_BENCH_STUB=1: no GPU here).
Also: the traffic record bench.py reports is dropped as stale when the kernel sources changed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _run(extra_env, *args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(S2A_BENCH_STUB="1", **extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True,
                          text=True, timeout=300)


def test_self_launch_two_ranks_gloo():
    r = _run({}, "--gpus", "2", "--steps", "1", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 1 and out["scaling"] == "weak" and out["stub"] is True
    assert out["config"]["global_batch"] == 16 and out["config"]["nms_candidates_dropped"] == 0
    assert out["gather_equals_concatenation"] is True      # gathered == rank-major concatenation of the rank outputs
    assert "roofline" not in out and "cpu_baseline" not in out
    # per-rank attribution for a future scaling loss: the timed step, the step without the collective, the collective alone
    assert [r["rank"] for r in out["per_rank"]] == [0, 1]
    assert all(r["compute_ms"] >= 0 and r["gather_ms"] >= 0 and r["step_ms"] > 0 for r in out["per_rank"])


def test_self_launch_eight_ranks_gloo():
    """the driver's N = 8 shape (BASELINE configs[3]: batch 64 over 8 ranks), rehearsed on CPU with the stub detector"""
    r = _run({}, "--gpus", "8", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["global_batch"] == 64 and out["stub"] is True
    assert out["gather_equals_concatenation"] is True and len(out["per_rank"]) == 8
    assert out["config"]["parallelism"].startswith("dp8")


def test_failed_rank_fails_the_launch():
    r = _run({"S2A_BENCH_STUB_FAIL_RANK": "1"}, "--gpus", "2", "--steps", "1", "--warmup", "1")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "rank 1 exited with 3" in r.stderr


def test_launcher_refuses_under_a_preloaded_profiler():
    """rocprofv3 preloads its tool library, which initialises the GPU before main(): spawning ranks from such a process
    is the hop that takes a box down.  bench.py must exit non-zero WITHOUT starting a child."""
    import bench
    assert bench.profiler_preloaded({"ROCP_TOOL_LIBRARIES": "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so"})
    assert bench.profiler_preloaded({"LD_PRELOAD": "/opt/rocm/lib/librocprofiler-sdk-tool.so.0"})
    assert not bench.profiler_preloaded({"LD_PRELOAD": "/lib/libfoo.so"}) and not bench.profiler_preloaded({})
    # HSA_TOOLS_LIB is only read when HSA initialises (never in this CPU process): safe to set for the subprocess
    r = _run({"HSA_TOOLS_LIB": "librocprofiler-sdk-tool.so"}, "--gpus", "2", "--steps", "1", "--warmup", "1")
    assert r.returncode == 4, (r.returncode, r.stderr[-500:])
    assert "refusing to launch" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_under_torchrun_env_no_relaunch():
    """WORLD_SIZE already set (torch.distributed.run): the process is a rank, it does not spawn"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, S2A_BENCH_STUB="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_recorded_traffic_goes_stale_with_the_sources(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    src = "s2anet_amd/csrc/dcn_ops.hip"
    rec = {"kernels": {"align_conv_pyramid": {"kernel": "k_dcn_patch", "fetch_kib": 100.0, "write_kib": 50.0,
                                              "bytes": (2 * 100.0 + 50.0) * 1024, "batch": 8, "pixels": 174592,
                                              "sources": [src], "source_sha16": bench._sha16([src])}}}
    f = tmp_path / "r99_traffic.json"
    f.write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "traffic_json", lambda: str(f))
    assert bench.recorded_traffic("align_conv_pyramid", 8, 174592) == (256000.0, False)
    assert bench.recorded_traffic("align_conv_pyramid", 4, 174592 // 2) == (None, True)      # another launch shape
    assert bench.recorded_traffic("conv_tower_pyramid", 8, 174592) == (None, True)           # no record
    rec["kernels"]["align_conv_pyramid"]["source_sha16"] = "0" * 16                           # kernel edited since
    f.write_text(json.dumps(rec))
    assert bench.recorded_traffic("align_conv_pyramid", 8, 174592) == (None, True)
    monkeypatch.setattr(bench, "traffic_json", lambda: None)
    assert bench.recorded_traffic("align_conv_pyramid", 8, 174592) == (None, True)
