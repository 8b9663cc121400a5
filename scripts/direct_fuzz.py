"""Randomised multi-rank runs of the direct paths on ONE GPU (companion of direct_stress.py): several rounds,
each with 2-5 ranks and three generated systems of random size and band (including bands wider than a shard, so
that a rank talks to more than its two neighbours, shards of a few dozen rows, and non-symmetric values); every
rank's slice of A.x (12 alternating calls) and of the solutions is compared with the single-process run.

  python scripts/direct_fuzz.py [rounds=6] [seed=1]
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    import test_gpu_direct as T
    for rnd in range(rounds):
        world = int(rng.integers(2, 6))
        cases = []
        for tag, sym in (("band", True), ("scr", True), ("nsym", False)):
            n = int(rng.integers(40 * world, 120000))
            if tag == "scr":
                band = 0
            else:
                band = int(rng.integers(1, max(2, n // 2)))
            cases.append((tag, n, band, sym))
        tmp = tempfile.mkdtemp()
        ref_path = os.path.join(tmp, "ref.npz")
        ref = T._reference(ref_path, tuple(cases))
        procs, outs = [], []
        for r in range(world):
            out = os.path.join(tmp, f"w{r}.json")
            outs.append(out)
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + rnd),
                       LCG_HIP_P2P_TIMEOUT_MS="8000")
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_direct_worker.py"), ref_path, out], env=env,
                                          stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
        errs = [p.communicate(timeout=600)[1] for p in procs]
        assert all(p.returncode == 0 for p in procs), (world, cases, [e[-1500:] for e in errs])
        res = [json.load(open(o)) for o in outs]
        worst = 0.0
        for r in res:
            for k, v in r.items():
                if k.endswith("spmv_err"):
                    worst = max(worst, v)
                    assert v < 1e-12, (world, cases, k, v)
                if isinstance(v, list):
                    assert v[0] == 0 or v[0] == -1019 or v[0] == 2, (world, cases, k, v)      # converged / cap / already optimal
        for key in res[0]:
            if isinstance(res[0][key], list):
                assert len({tuple(r[key][:2]) for r in res}) == 1, (world, cases, key)
        print(f"round {rnd}: world {world} cases {cases}: worst A.x error {worst:.1e}", flush=True)
    print("direct fuzz: ok")


if __name__ == "__main__":
    main()
