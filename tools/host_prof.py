import os, sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import torch
import artensor_amd as A
from artensor_amd.fixtures import load_case
case = load_case("/root/repo/tests/golden/n12_dense.npz")
leaves = case.fresh_tensors(device="cuda")
for _ in range(5): A.tensor_contraction(dict(leaves), case.scheme)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300): A.tensor_contraction(dict(leaves), case.scheme)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
