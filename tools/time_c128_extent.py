import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import artensor_amd as A
from artensor_amd.fixtures import load_case
for name in ("rand_D3_nv112", "rand_D6_nv64"):
    case = load_case(f"/root/repo/tests/golden/{name}.npz")
    for dt in (torch.complex64, torch.complex128):
        t = case.fresh_tensors(dtype=dt, device="cuda")
        A.tensor_contraction(dict(t), case.scheme); torch.cuda.synchronize()
        t0 = time.perf_counter(); out = A.tensor_contraction(dict(t), case.scheme); torch.cuda.synchronize(); dt_s = time.perf_counter() - t0
        ref = case.arrays.get("exact128")
        err = abs(complex(out.cpu().reshape(-1)[0]) - complex(ref.reshape(-1)[0])) / abs(complex(ref.reshape(-1)[0])) if ref is not None and out.numel() == 1 else float("nan")
        print(name, dt, f"{dt_s*1e3:.1f} ms", f"rel err vs reference complex128 {err:.2e}", flush=True)
