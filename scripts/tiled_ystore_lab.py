"""LAB build only (make -C liblcg_amd/csrc LAB=1; LCG_HIP_LAB=1): does the cache policy of the tiled product's y stores change its
slow state?  One process, one placement (LCG_HIP_PLACE=0: the vectors stay as allocated), the policy switched between solves
(LCG_HIP_Y_STORE is read at every launch): 0 plain, 1 non-temporal, 2 sc1, 3 sc0 sc1.  A.x in the CG loop, HIP events.
Needs the tiled kernel's epilogue to store through store_y(..., dp.ystore) (a one-line LAB patch, not in the tree).  Result (round 5,
profiles/r05_tiled_states.txt): no policy moves the product by more than 0.1 %."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LCG_HIP_LAB", "1")
import torch
from liblcg_amd import _lib, api
lib = _lib.load()
assert "lab" in _lib.SO_PATH, _lib.SO_PATH
lib.lcg_hip_set_placement(int(os.environ.get("PLACE", "0")))
N = 10000000
A = api.CsrMatrix.generate(N, 16, 131072, True, 1, 0.01, pattern=api.GEN_ROW_RANDOM_BAND)
xt = torch.empty(N, dtype=torch.float64, device="cuda"); api.gen_xtrue(N, 1, 0, N, xt)
b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
m = torch.zeros_like(xt)
p = api.lcg_default_parameters(epsilon=1e-300, max_iterations=60)
api.lcg("lcg_hip_csr_ax", None, m, b, N, p, A); api.synchronize()
print(lib.lcg_hip_csr_last_kernel(A.h).decode()[:60])
for rep in range(3):
    row = []
    for pol in (0, 1, 2, 3):
        os.environ["LCG_HIP_Y_STORE"] = str(pol)
        m.zero_(); lib.lcg_hip_set_profiling(1)
        api.lcg("lcg_hip_csr_ax", None, m, b, N, p, A); api.synchronize()
        row.append(lib.lcg_hip_last_ax_mean_us())
    print("rep", rep, " ".join(f"policy {k}: {v:6.1f} us" for k, v in enumerate(row)), flush=True)
