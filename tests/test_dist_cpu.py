"""world_size-2 gloo test (CPU) of the multi-GPU plumbing: contiguous batch sharding by rank and the
single all-gather of packed detections [B_local, Q, C+4] (dinov2_od_amd/dist.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dinov2_od_amd.dist import shard_bounds


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, global_batch, ragged, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from dinov2_od_amd import dist as ddist
    ddist.init_from_env("gloo")
    lo, hi = shard_bounds(global_batch, rank, world)
    Q, C = 5, 7
    full = torch.arange(global_batch * Q * (C + 4), dtype=torch.float32).view(global_batch, Q, C + 4)
    local = full[lo:hi].clone()                  # stands for this rank's forward_packed() output
    out = ddist.gather_detections(local) if ragged else ddist.gather_detections_equal(local)
    ok = torch.equal(out, full)
    q.put((rank, ok, tuple(out.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("global_batch,ragged", [(8, False), (8, True), (7, True)])
def test_gather_detections_world2(global_batch, ragged):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, global_batch, ragged, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, shape in res:
        assert ok, (rank, shape)
        assert shape[0] == global_batch


def test_shard_bounds_cover_batch():
    for gb in (1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_bench_self_launches_ranks_and_gathers(tmp_path):
    """`python bench.py --gpus 2` with no torchrun environment: the parent spawns two fresh rank processes (dinov2_od_amd/launch.py,
    the reference's mp.spawn at train.py:1501-1506), they rendezvous on 127.0.0.1 over gloo, run the sharded step loop with the
    gather and rank 0 prints the one JSON line.  --rehearse-cpu swaps the model for a stub: control flow only."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse-cpu"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["n_ranks_seen"] == 2 and j["config"]["global_batch"] == 128
    assert j["config"]["gather_checked"] is True
    assert j["data"] == "rehearsal-cpu" and j["metric"].startswith("REHEARSAL")
    assert j["global64_sharded"]["per_gpu_batch"] == 32 and j["global64_sharded"]["scaling"] == "strong"
    assert j["steps"] == 3 and j["value"] > 0
    # what the first unattended 8-GPU run must show: every rank's own clock and every rank's detections checked against the oracle
    pr = j["per_rank"]
    assert [r_["rank"] for r_ in pr["ms_per_step"]] == [0, 1] and all(r_["ms_per_step"] > 0 for r_ in pr["ms_per_step"])
    assert pr["spread_ms"] >= 0 and pr["slowest_rank"] in (0, 1)
    chk = pr["gpu_vs_oracle"]
    assert chk["all_in_gate"] and [r_["global_image"] for r_ in chk["ranks"]] == [0, 64]


def test_bench_under_the_drivers_torchrun_command_four_ranks(tmp_path):
    """the launch the driver uses at round end -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- with N = 4 (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment, no self-launch):
    one JSON line from rank 0, four ranks seen, every rank's shard timed and checked"""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
                        "--rehearse-cpu", "--no-extras"], capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 4 and j["config"]["n_ranks_seen"] == 4 and j["config"]["global_batch"] == 256 and j["scaling"] == "weak"
    pr = j["per_rank"]
    assert [r_["rank"] for r_ in pr["ms_per_step"]] == [0, 1, 2, 3]
    assert pr["gpu_vs_oracle"]["all_in_gate"] and [r_["global_image"] for r_ in pr["gpu_vs_oracle"]["ranks"]] == [0, 64, 128, 192]


def test_bench_fails_when_a_rank_leaves_the_gate(tmp_path):
    """a rank whose detections are wrong must fail the JOB (non-zero exit of `bench.py --gpus N`), not only show up in the JSON"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_REHEARSE_BREAK_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rehearse-cpu", "--no-extras"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r.returncode != 0
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    ranks = j["per_rank"]["gpu_vs_oracle"]["ranks"]
    assert ranks[0]["in_gate"] and not ranks[1]["in_gate"] and not j["per_rank"]["gpu_vs_oracle"]["all_in_gate"]


def test_launcher_propagates_a_failing_rank(tmp_path):
    from dinov2_od_amd.launch import spawn_ranks
    import io
    script = tmp_path / "child.py"
    script.write_text("import os, sys, time\n"
                      "r = int(os.environ['RANK'])\n"
                      "assert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                      "print('hello from', r, flush=True)\n"
                      "if r == 1: sys.exit(7)\n"
                      "time.sleep(30)\n")          # rank 0 would wait in a collective: the launcher must end it
    out, err = io.StringIO(), io.StringIO()
    t0 = __import__("time").time()
    rc = spawn_ranks(str(script), [], 2, timeout=60, stdout=out, stderr=err)
    assert rc == 7 and __import__("time").time() - t0 < 25
    assert "hello from 0" in out.getvalue() and "hello from 1" in err.getvalue()
