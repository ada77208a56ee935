"""bench.py's own launcher, step loop protocol (warm-up, barrier + synchronize on both sides, max over ranks), GradSync wiring and
JSON line at world size 2 - on CPU over gloo with `--dry-run` (a stand-in for the model; the HIP path itself needs the GPU).
VERDICT r2 #2: `python bench.py --gpus N` must launch its own workers, and the same file must work under the driver's
`python -m torch.distributed.run ... bench.py --gpus N` form."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_lines(out):
    rows = []
    for ln in out.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                rows.append(json.loads(ln))
            except ValueError:
                pass
    return rows


def _check(rows, n):
    assert len(rows) == 1, rows  # ONE line, from rank 0
    r = rows[0]
    assert r["n_gpus"] == n and r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "weak" and r["value"] > 0
    assert abs(r["value"] - n * r["tokens_per_s_per_gpu"]) < 1.0  # whole-job aggregate
    assert r["config"]["parallelism"] == f"dp{n}" and r["dry_run"] is True
    if n > 1:
        assert r["rccl_ranks"] == n and len(r["per_rank"]) == n
        assert r["comm_collectives_per_step"] == 5  # one all-reduce per bucket of the stand-in engine, every step
        assert r["comm_ms_total"] >= r["comm_ms_exposed"] >= 0 and r["recompute_fallback"] is False


def test_bench_self_launches_two_workers_and_prints_one_line():
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    _check(_json_lines(p.stdout), 2)


def test_bench_self_launches_eight_workers_the_scale_run_shape():
    """The driver's SCALE run ends at N = 8 (one rank per GPU of the node): the launcher, the rendezvous, GradSync's fixed collective order, the
    per-rank gathers and the ONE line, with eight gloo workers on this host."""
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    _check(_json_lines(p.stdout), 8)


def test_bench_under_the_drivers_torchrun_command_line():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    _check(_json_lines(p.stdout), 2)


def test_bench_single_process_dry_run_and_mismatch_is_an_error():
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    _check(_json_lines(p.stdout), 1)
    env["WORLD_SIZE"] = "4"  # a torchrun environment that disagrees with --gpus
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode != 0 and "nproc-per-node" in (p.stderr + p.stdout)


def test_committed_pmc_traffic_summary_was_taken_on_the_committed_gemm_sources():
    """the driver's bench line carried `roofline.traffic: null` because a GEMM source changed after the PMC passes.  The
    newest profiles/rNN_gemm_traffic.json must carry the hash of the GEMM kernel sources as committed: the PMC pass
    (tools/pmc_step_traffic.sh) is the LAST GPU action after any change to gemm*.hip / gemm_common.h / mh_common.h."""
    sys.path.insert(0, ROOT)
    import bench

    path = bench.traffic_summary_path()
    assert path is not None, "no profiles/rNN_gemm_traffic.json"
    with open(path) as f:
        t = json.load(f)
    assert t["kernel_source_stamp"] == bench.kernel_source_stamp(), (os.path.basename(path), t["kernel_source_stamp"], bench.kernel_source_stamp())
    assert t["traffic_over_algorithmic"] > 1.0 and t["traffic_bytes_per_launch"] > 0


def test_skinny_split_k_is_a_bf16_launch_plan_and_the_same_inside_a_graph_capture(monkeypatch):
    """ops._skinny_splitk_ok: split-K of the skinny projections applies to bf16 (the performance dtype); fp16 - the dtype BASELINE's logits
    tolerance is stated in - keeps the one-pass summation order its full-depth bounds were measured with.  The decision is the SAME while a
    stream is being captured into a HIP graph (graph and eager logits bit-identical): the partials' workspace then comes from
    the graph's own pool and is not kept in the per-stream cache (ops._splitk_workspace)."""
    import torch

    sys.path.insert(0, ROOT)
    from merlin_amd import ops as O

    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    monkeypatch.setattr(O, "SKINNY_SPLITK", True)
    monkeypatch.setattr(O, "SKINNY_SPLITK_ALL", False)
    assert O._skinny_splitk_ok(torch.bfloat16) and not O._skinny_splitk_ok(torch.float16)
    monkeypatch.setattr(O, "SKINNY_SPLITK_ALL", True)
    assert O._skinny_splitk_ok(torch.float16)
    monkeypatch.setattr(O, "SKINNY_SPLITK_ALL", False)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: True)
    assert O._skinny_splitk_ok(torch.bfloat16) and not O._skinny_splitk_ok(torch.float16)
    # during a capture the workspace is a fresh allocation (the capture's pool) and the per-stream cache is left alone
    made = []
    monkeypatch.setattr(torch, "empty", lambda *a, **k: made.append((a, k)) or "ws")
    before = dict(O._splitk_ws)
    assert O._splitk_workspace(torch.device("cpu"), 1234) == "ws" and made and made[0][0] == (1234,) and O._splitk_ws == before
    monkeypatch.setattr(O, "SKINNY_SPLITK", False)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    assert not O._skinny_splitk_ok(torch.bfloat16)
