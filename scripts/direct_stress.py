"""Stress of the multi-rank direct paths on ONE GPU at sizes near the real shards (not a pytest test:
minutes, GBs).  RANKS processes share GPU 0 (no RCCL: mailboxes + direct exchange), every rank's
slice of A.x and of the CG / BiCGStab / CGS solutions is compared with the single-process run --
the checks of tests/test_gpu_direct.py on a 3M-row banded system with the headline band and on a
1M-row scrambled one, i.e. hundreds of iterations with 2 MB pushes per call.

  python scripts/direct_stress.py [RANKS=4]
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    import test_gpu_direct as T
    tmp = tempfile.mkdtemp()
    ref_path = os.path.join(tmp, "ref.npz")
    cases = (("band", 3_000_000, 131072, True), ("scr", 1_000_003, 0, True), ("nsym", 2_000_000, 50_000, False))
    if os.environ.get("LCG_STRESS_CASES"):
        cases = tuple(c for c in cases if c[0] in os.environ["LCG_STRESS_CASES"].split(","))
    ref = T._reference(ref_path, cases)
    procs, outs = [], []
    for r in range(world):
        out = os.path.join(tmp, f"w{r}.json")
        outs.append(out)
        env = dict(os.environ, LCG_DIRECT_VERBOSE="1", RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29581")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_direct_worker.py"), ref_path, out], env=env))
    rc = [p.wait(timeout=1500) for p in procs]
    assert all(c == 0 for c in rc), rc
    res = [json.load(open(o)) for o in outs]
    for r in res:
        print(json.dumps(r))
        for k, v in r.items():
            if k.endswith("spmv_err"):
                assert v < 1e-13, (k, v)
            if isinstance(v, list):
                assert v[0] == 0, (k, v)
    for key in res[0]:
        if isinstance(res[0][key], list):
            assert len({tuple(r[key][:2]) for r in res}) == 1, key
    for tag, _, _, sym in cases:
        for name in ("cg", "cgs"):
            if f"{tag}/{name}" in res[0]:
                print(tag, name, "iterations sharded", res[0][f"{tag}/{name}"][1], "single", int(ref[f"{tag}/{name}_its"]))
                assert abs(res[0][f"{tag}/{name}"][1] - int(ref[f"{tag}/{name}_its"])) <= 3
    print("direct stress: ok")


if __name__ == "__main__":
    main()
