"""Child process of tests/test_gpu_parity.py::test_two_process_sliced_contraction: one rank of a
world_size-2 group (gloo rendezvous on 127.0.0.1; both ranks on cuda:0, which RCCL cannot do but the
product code path does not care) running the PRODUCT slice loop -- real HIP executors, real
accumulate, one all_reduce -- and writing its result for the parent to compare.

    python tests/dist_gpu_worker.py RANK WORLD PORT OUT_DIR [BACKEND [DEVICE]]

BACKEND nccl (= RCCL; one GPU per rank: DEVICE defaults to RANK) is what a multi-GPU node runs
(test_two_gpu_rccl_sliced_contraction, skipped on one-GPU boxes).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
backend = sys.argv[5] if len(sys.argv) > 5 else "gloo"
device = int(sys.argv[6]) if len(sys.argv) > 6 else (rank if backend == "nccl" else 0)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
if backend == "nccl":
    torch.cuda.set_device(device)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
else:
    dist.init_process_group("gloo", rank=rank, world_size=world)
import artensor_amd as A
from artensor_amd.fixtures import load_case

GOLDEN = os.path.join(ROOT, "tests", "golden")
dev = f"cuda:{device}"
try:
    # (1) a whole sliced contraction, sharded over the two ranks, all_reduce at the end
    case = load_case(os.path.join(GOLDEN, "n12_sparse_sliced.npz"))
    want = case.arrays["final"]
    out = A.sliced_contraction(case.tensors, case.scheme, case.slicing_indices, want.shape, sparse=True, device=dev)
    np.save(os.path.join(out_dir, f"n12_{rank}.npy"), out.cpu().numpy())
    # (2) 8 slices of n53 m14 (big steps on the MFMA kernels), explicit slice lists, reduce to rank 0 only
    case = load_case(os.path.join(GOLDEN, "n53_m14_sliced.npz"))
    mine = [s for s in range(8) if s % world == rank]
    out = A.sliced_contraction(case.fresh_tensors(device=dev), case.scheme, case.slicing_indices, (1,), sparse=True,
                               device=dev, slices=mine, reduce="root")
    np.save(os.path.join(out_dir, f"n53_{rank}.npy"), out.cpu().numpy())
    # (3) the collective really ran over both ranks' GPU buffers (gloo stages them through the host)
    t = torch.full((4,), float(rank + 1), device=dev)
    dist.all_reduce(t)
    assert float(t[0].item()) == sum(range(1, world + 1))
finally:
    dist.destroy_process_group()
