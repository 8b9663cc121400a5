"""`python bench.py --gpus N` must produce its line by itself: the launcher spawns N fresh ranks under
torch.distributed.run before anything touches the GPU and relays rank 0's one JSON line.  Rehearsed here on CPUs with
--dry-launch (gloo; every rank reports the environment the launcher gave it), including a rank that dies and ranks that never exit."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(cmd, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LCG_BENCH_T0"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    return p.returncode, lines, p.stderr


def test_plain_command_launches_its_own_ranks():
    rc, lines, err = _run([sys.executable, BENCH, "--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-launch"])
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines                    # ONE relayed line on the standard output, nothing else
    d = json.loads(lines[0])
    assert d["dry_launch"] and d["n_gpus"] == 2 and d["gpus_asked"] == 2
    assert d["ranks"] == [0, 1] and d["local_ranks"] == [0, 1] and d["world_sizes"] == [2, 2]
    assert len(set(d["pids"])) == 2 and os.getpid() not in d["pids"]      # fresh child processes
    assert d["master"].startswith("127.0.0.1:")
    assert (d["steps"], d["warmup"]) == (20, 5)      # the same arguments reached the ranks
    assert "launching 2 ranks" in err


def test_three_ranks():
    rc, lines, err = _run([sys.executable, BENCH, "--gpus", "3", "--dry-launch"])
    assert rc == 0, err[-2000:]
    d = json.loads(lines[-1])
    assert d["ranks"] == [0, 1, 2] and d["world_sizes"] == [3, 3, 3]


def test_driver_form_under_torch_distributed_run():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    rc, lines, err = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-launch"])
    assert rc == 0, err[-2000:]
    d = json.loads([ln for ln in lines if ln.startswith("{")][-1])
    assert d["ranks"] == [0, 1] and "launching" not in err      # inside a launcher's environment nothing is spawned again


def test_a_rank_that_dies_still_leaves_one_line_and_a_nonzero_code():
    rc, lines, err = _run([sys.executable, BENCH, "--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-launch"], {"LCG_BENCH_DRY_FAIL": "1"})
    assert rc != 0
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] == 0.0 and "error" in d and d["n_gpus"] == 2 and d["metric"] == "cg_iterations_per_sec"
    assert (d["steps"], d["warmup"]) == (20, 5)


def test_ranks_that_never_exit_are_ended_and_the_result_kept():
    t0 = time.time()
    rc, lines, err = _run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], {"LCG_BENCH_DRY_SLEEP": "600", "LCG_BENCH_EXIT_GRACE": "3"})
    assert time.time() - t0 < 120
    assert rc == 0 and len(lines) == 1
    d = json.loads(lines[0])
    assert d["ranks"] == [0, 1] and "ended by the launcher" in d["launcher_note"]
    for pid in d["pids"]:                            # the launcher's own session is gone
        for _ in range(50):
            if not os.path.exists(f"/proc/{pid}"):
                break
            time.sleep(0.1)
        assert not os.path.exists(f"/proc/{pid}")


def test_no_result_inside_the_budget_is_an_error_line():
    rc, lines, err = _run([sys.executable, BENCH, "--gpus", "2", "--dry-launch", "--budget-seconds", "-40"],
                          {"LCG_BENCH_DRY_SLEEP": "600", "LCG_BENCH_DRY_FAIL": "-1", "LCG_BENCH_EXIT_GRACE": "600"})
    # the ranks DID report here (a dry launch always does): the line is kept and flagged; what matters is that the launcher came back
    assert len(lines) == 1 and "launcher_note" in json.loads(lines[0])
