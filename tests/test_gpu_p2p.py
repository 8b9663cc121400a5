"""-m gpu: the direct all-reduce (include/lcg_hip.h: lcg_hip_p2p_*) as far as ONE GPU can show it.

 * three processes on GPU 0 map each other's mailboxes through HIP IPC (the same calls a node
   with one GPU per process makes), pass the self-test and all-reduce random values: every
   result equals the rank-ordered sum bit for bit, on every rank;
 * a one-rank run (P = 1: the mailbox is the rank's own) drives the fused reduce + exchange +
   scalar-step kernel through every solver family and reproduces the RCCL-path results;
 * a peer that never shows up ends the call after the timeout instead of hanging.
What one GPU cannot show -- coherence of the polled mailbox against writes that arrive over the
fabric -- is what the self-test at start-up decides on the real node (partition.init_p2p_from_torch).
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_three_processes_on_one_gpu(tmp_path):
    world = 3
    procs, outs = [], []
    for r in range(world):
        out = str(tmp_path / f"p2p_{r}.json")
        outs.append(out)
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29561")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_p2p_worker.py"), out],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    logs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(so[-1500:] + se[-3000:])
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    res = [json.load(open(o)) for o in outs]
    assert all(r["enabled"] for r in res), res
    assert all(r["status"] == 2 for r in res)
    assert all(r["calls"] == 60 and r["mismatch"] == 0 for r in res), res


def test_missing_peer_times_out():
    """Rank 0 of 2 whose peer exported a mailbox but never takes part: the self-test gives up after
    the timeout and reports LCG_HIP_E_COMM; nothing hangs."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_p2p_worker.py"), "--lonely"],
                       capture_output=True, text=True, timeout=200)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "RC -2002" in p.stdout


def _solve_cases(tmp_path, tag, extra_env):
    out = str(tmp_path / f"{tag}.npz")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_sharded_case.py"), out],
                       capture_output=True, text=True, env=dict(os.environ, **extra_env), timeout=280)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return np.load(out)


def test_fused_exchange_step_reproduces_the_rccl_path(tmp_path):
    """One rank, forced communicator: with the direct all-reduce every sync point is ONE kernel
    (reduce + mailbox exchange + scalar step); with P = 1 its sums are the rank's own, so every
    solver must return exactly what the reduce | ncclAllReduce | scalar-step path returns."""
    rccl = _solve_cases(tmp_path, "rccl", {"LCG_HIP_FORCE_COMM": "1", "LCG_HIP_P2P": "0", "MASTER_PORT": "29562"})
    direct = _solve_cases(tmp_path, "direct", {"LCG_HIP_FORCE_COMM": "1", "LCG_HIP_P2P": "1", "MASTER_PORT": "29563"})
    assert int(direct["p2p_status"]) == 2 and int(rccl["p2p_status"]) == 0
    for key in rccl.files:
        if key == "p2p_status":
            continue
        assert np.array_equal(rccl[key], direct[key]), key
