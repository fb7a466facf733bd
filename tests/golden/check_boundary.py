#!/usr/bin/env python3
"""Validate the drop-in boundary on LIVE reference objects (build container only, CPU only).

    PYTHONHASHSEED=0 PYTHONDONTWRITEBYTECODE=1 python tests/golden/check_boundary.py

Imports the reference (/root/reference) and this package side by side, plans the bundled n12 m14
circuit three ways with the reference's planner (dense, sparse, sparse + sliced) and checks that

  1. artensor_amd.contraction_scheme / contraction_scheme_sparse, fed a live ContractionTree
     (contraction_tree.py:117-134, mark_rep_tensor :305-314, tree_order_dfs :334-357), return exactly
     what the reference's compilers return (contraction.py:23-59, :208-341): equations, edges, index
     tensors, reshape shapes, bitstrings_sorted;
  2. TensorNetworkSimulation.from_planned(sim) picks up the attributes the reference's
     prepare_contraction / update_scheme leave behind (simulation.py:47-88), by their own names;
  3. the four executor / compiler functions and the two one-call entry points keep the reference's
     parameter names and defaults (inspect.signature; this package may append keyword parameters);
  4. artensor_amd.TensorNetworkSimulation.from_circuit_file(...).prepare_contraction(...) and
     artensor_amd.quantum_circuit_simulation(...) -- the planner-forwarding halves of the boundary --
     reproduce the reference's scheme / slicing / permutation and, with the CPU oracle standing in for
     the HIP executors (there is no GPU here), the reference's amplitudes.

Writes the pass record to tests/golden/boundary_check.json, which tests/test_boundary_record.py asserts
on every box.  Exits non-zero on the first failed check.  No reference source is stored: the record
holds check names, counts and error figures only.
"""
import inspect
import json
import os
import sys
import time
from copy import deepcopy

if os.environ.get("PYTHONHASHSEED") != "0":
    sys.exit("run with PYTHONHASHSEED=0 (schemes depend on the str hash seed)")
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import artensor as R          # the reference  # noqa: E402
import artensor_amd as A      # this package   # noqa: E402
from artensor_amd import simulation as S  # noqa: E402
from oracle import oracle     # CPU stand-in for the HIP executors (test infrastructure)  # noqa: E402

N12 = os.path.join(REF, "tests", "circuit_n12_m14_s0_e0_pEFGH.qsim")
PLAN = dict(trials=5, iters=10, slicing_repeat=1)
record = {"checks": [], "torch": torch.__version__, "hashseed": 0}


def check(name, ok, **info):
    record["checks"].append(dict(name=name, passed=bool(ok), **info))
    print(("ok   " if ok else "FAIL ") + name, info, flush=True)
    if not ok:
        sys.exit(f"boundary check failed: {name}")


def same_sparse_scheme(mine, ref):
    if len(mine) != len(ref):
        return False
    for a, b in zip(mine, ref):
        if len(a) != len(b) or tuple(a[0]) != tuple(b[0]) or a[1] != b[1]:
            return False
        for side in (0, 1):
            if len(a[2][side]) != len(b[2][side]):
                return False
            for x, y in zip(a[2][side], b[2][side]):
                if not torch.equal(torch.as_tensor(x), torch.as_tensor(y)):
                    return False
        if len(a) == 5 and (a[3] != b[3] or tuple(a[4]) != tuple(b[4])):
            return False
    return True


def plan(bitstrings, sc_target):
    sim = R.TensorNetworkSimulation.from_circuit_file(N12, bitstrings)
    sim.prepare_contraction(sc_target=sc_target, **PLAN)
    return sim


rng = np.random.RandomState(3)
bits40 = sorted({np.binary_repr(x, 12) for x in rng.randint(0, 4096, size=40)})
rng = np.random.RandomState(7)
bits20 = sorted({np.binary_repr(x, 12) for x in rng.randint(0, 4096, size=24)})[:20]

# ---- 1. compilers on live trees ---------------------------------------------------------------------
dense = plan([], 30)
mine, out_m = A.contraction_scheme(deepcopy(dense.ctree))
ref, out_r = R.contraction_scheme(deepcopy(dense.ctree))
check("contraction_scheme(live ContractionTree) == reference", mine == ref and list(out_m) == list(out_r), steps=len(ref))

sparse = plan(bits40, 30)
for sc in (30, 8, 6):
    m = A.contraction_scheme_sparse(deepcopy(sparse.ctree), bits40, sc_target=sc)
    r = R.contraction_scheme_sparse(deepcopy(sparse.ctree), bits40, sc_target=sc)
    check(f"contraction_scheme_sparse(live tree, sc_target={sc}) == reference",
          same_sparse_scheme(m[0], r[0]) and list(m[1]) == list(r[1]) and list(m[2]) == list(r[2]),
          steps=len(r[0]), five_tuples=sum(len(s) == 5 for s in r[0]), chunked=sum(len(s[2][0]) > 1 for s in r[0]))

# a big batch on the same tree: 1 500 random bitstrings, every branch of the compiler (outer products with and without a
# row select, chunked gathers of several sizes) -- the vectorised compiler against the reference's bitstring-by-bitstring loops
rng = np.random.RandomState(11)
bits_many = sorted({np.binary_repr(x, 12) for x in rng.randint(0, 4096, size=2200)})[:1500]
big = plan(bits_many, 30)
for sc in (30, 12, 9):
    m = A.contraction_scheme_sparse(deepcopy(big.ctree), bits_many, sc_target=sc)
    r = R.contraction_scheme_sparse(deepcopy(big.ctree), bits_many, sc_target=sc)
    check(f"contraction_scheme_sparse(1 500 bitstrings, sc_target={sc}) == reference",
          same_sparse_scheme(m[0], r[0]) and list(m[1]) == list(r[1]) and list(m[2]) == list(r[2]),
          steps=len(r[0]), five_tuples=sum(len(s) == 5 for s in r[0]), chunked=sum(len(s[2][0]) > 1 for s in r[0]),
          selects=sum(len(s) == 5 and len(s[2][0]) == 1 and len(s[2][1]) == 0 for s in r[0]))


# the n30 m14 tree with Google's 10 000 bitstrings (BASELINE configs[2], SURVEY 8a row A4: the reference's compiler needs
# 10-40 s for it): the first live equality above 1 500 rows, on the circuit and batch the bench leg runs
import make_golden as MG  # noqa: E402  (same directory: the n30 .qsim conversion and the amplitude file reader)
t0 = time.time()
g_bits, _ = MG.read_google(10000)
n30 = R.TensorNetworkSimulation.from_circuit_file(MG.n30_qsim(), g_bits)
n30.prepare_contraction(sc_target=30, **PLAN)
t_plan = time.time() - t0
t0 = time.time()
r = R.contraction_scheme_sparse(deepcopy(n30.ctree), g_bits, sc_target=30)
t_ref = time.time() - t0
t0 = time.time()
m = A.contraction_scheme_sparse(deepcopy(n30.ctree), g_bits, sc_target=30)
t_mine = time.time() - t0
check("contraction_scheme_sparse(n30 m14 tree, Google's 10 000 bitstrings) == reference",
      same_sparse_scheme(m[0], r[0]) and list(m[1]) == list(r[1]) and list(m[2]) == list(r[2]),
      steps=len(r[0]), five_tuples=sum(len(s) == 5 for s in r[0]), chunked=sum(len(s[2][0]) > 1 for s in r[0]),
      rows=len(r[2]), seconds_reference=round(t_ref, 1), seconds_here=round(t_mine, 2), seconds_planning=round(t_plan, 1))
del n30, r, m


def slicing_ok(sim):
    """True when the reference's slice loop is well defined for this plan: no tensor
    carries two sliced bonds in ascending dim order, and no sliced bond sits on a leaf whose leading
    batch dim tensor_bonds does not list (simulation.py:62-65 would then select() the batch dim)."""
    seen = {}
    for bond, lst in sim.slicing_indices.items():
        for tid, ind in lst:
            if sim.tensors[tid].dim() != len(sim.tensor_bonds[tid]):
                return False
            if any(prev < ind for prev in seen.get(tid, [])):
                return False
            seen.setdefault(tid, []).append(ind)
    return True


sliced = None
for sc_target in (10, 9, 11, 8, 7):
    cand = plan(bits20, sc_target)
    if len(cand.slicing_indices) >= 2 and slicing_ok(cand):
        sliced = cand
        break
check("a sliced sparse plan exists", sliced is not None, sc_target=sc_target,
      sliced_bonds=len(sliced.slicing_indices) if sliced else 0)
m = A.contraction_scheme_sparse(deepcopy(sliced.ctree), bits20, sc_target=sc_target)
check("contraction_scheme_sparse(sliced live tree) == reference scheme held by the planned object",
      same_sparse_scheme(m[0], sliced.scheme) and list(m[2]) == list(sliced.bitstrings_sorted), steps=len(sliced.scheme))

# ---- 2. from_planned ----------------------------------------------------------------------------------
for name, sim in (("dense", dense), ("sparse", sparse), ("sliced", sliced)):
    mine = A.TensorNetworkSimulation.from_planned(sim)
    ok = (mine.scheme is sim.scheme and mine.tensors is sim.tensors and mine.pattern == sim.pattern
          and list(mine.slicing_indices.keys()) == list(sim.slicing_indices.keys())
          and mine.output_bonds == list(sim.output_bonds)
          and getattr(mine, "bitstrings_sorted", None) == getattr(sim, "bitstrings_sorted", None)
          and list(getattr(mine, "permute_dims", []) or []) == list(getattr(sim, "permute_dims", []) or []))
    check(f"from_planned({name}) picks up scheme / tensors / slicing_indices / output_bonds / bitstrings_sorted / permute_dims",
          ok, attributes=sorted(k for k in vars(sim) if not k.startswith("_")))

# ---- 3. signatures ------------------------------------------------------------------------------------
for fn in ("contraction_scheme", "tensor_contraction", "contraction_scheme_sparse", "tensor_contraction_sparse",
           "tensor_network_contraction", "quantum_circuit_simulation"):
    pr = list(inspect.signature(getattr(R, fn)).parameters.values())
    pm = list(inspect.signature(getattr(A, fn)).parameters.values())
    same = len(pm) >= len(pr)
    for a, b in zip(pm, pr):
        default_ok = a.default == b.default or (a.name == "device" and b.default == "cpu")  # no CPU path here
        same = same and a.name == b.name and default_ok
    extra = [p.name for p in pm[len(pr):]]
    same = same and all(p.default is not inspect.Parameter.empty for p in pm[len(pr):])
    check(f"signature of {fn}", same, reference=[p.name for p in pr], appended_keywords=extra)
for meth in ("from_circuit_file", "from_tn_circuit", "prepare_contraction", "update_scheme", "contraction"):
    pr = [p for p in inspect.signature(getattr(R.TensorNetworkSimulation, meth)).parameters]
    pm = [p for p in inspect.signature(getattr(A.TensorNetworkSimulation, meth)).parameters]
    lead = pm[:len(pr)] == pr or (meth == "prepare_contraction" and pm[:2] == pr[:2])  # planner kwargs pass through
    check(f"TensorNetworkSimulation.{meth} parameters", lead, reference=pr, here=pm)

# ---- 4. planner-forwarding entry points, CPU oracle as executor ------------------------------------------
def _np(t):
    return t.numpy() if isinstance(t, torch.Tensor) else t


def oracle_dense(tensors, scheme):
    items = tensors.items() if isinstance(tensors, dict) else enumerate(tensors)
    return torch.from_numpy(np.ascontiguousarray(oracle.tensor_contraction({k: _np(v) for k, v in items}, scheme)))


def oracle_sparse(tensors, scheme, scientific_notation=False):
    items = tensors.items() if isinstance(tensors, dict) else enumerate(tensors)
    return torch.from_numpy(np.ascontiguousarray(oracle.tensor_contraction_sparse({k: _np(v) for k, v in items}, scheme)))


def cpu_add(acc, x):
    acc += x
    return acc


S.tensor_contraction, S.tensor_contraction_sparse, S.accumulate = oracle_dense, oracle_sparse, cpu_add   # no GPU here

for name, bits, sc in (("dense", [], 30), ("sparse", bits40, 30), ("sliced", bits20, sc_target)):
    ref_sim = {"dense": dense, "sparse": sparse, "sliced": sliced}[name]
    mine = A.TensorNetworkSimulation.from_circuit_file(N12, bits)
    mine.prepare_contraction(sc_target=sc, **PLAN)
    same = (mine.scheme == ref_sim.scheme) if name == "dense" else same_sparse_scheme(mine.scheme, ref_sim.scheme)
    same = same and list(mine.slicing_indices.keys()) == list(ref_sim.slicing_indices.keys())
    same = same and list(getattr(mine, "permute_dims", []) or []) == list(getattr(ref_sim, "permute_dims", []) or [])
    check(f"artensor_amd.TensorNetworkSimulation.from_circuit_file + prepare_contraction ({name}) == reference plan", same,
          steps=len(mine.scheme), sliced_bonds=len(mine.slicing_indices))
    want = ref_sim.contraction().reshape(-1).numpy()
    got = mine.contraction(device="cpu").reshape(-1).numpy()
    if name == "sliced":   # an independent check of the sliced sum: the reference's own state-vector simulator
        sv = R.TensorNetworkCircuit(N12).state_vec().reshape(-1)
        at = np.array([sv[int(b, 2)].item() for b in mine.bitstrings_sorted])
        check("sliced sum == TensorNetworkCircuit.state_vec() at the bitstrings", np.abs(got - at).max() < 2e-5 * np.abs(at).max(),
              rel_err=float(np.abs(got - at).max() / np.abs(at).max()))
    err = float(np.abs(got - want).max() / np.abs(want).max())
    check(f"... and its contraction() reproduces the reference's amplitudes ({name}; oracle as executor)", err < 1e-5, rel_err=err)

t0 = time.time()
want, wbits = R.quantum_circuit_simulation(N12, bits40, 30, trial_num=2)
got, gbits = A.quantum_circuit_simulation(N12, bits40, 30, trial_num=2, device="cpu")
err = float(np.abs(got.reshape(-1).numpy() - want.reshape(-1).numpy()).max() / np.abs(want.numpy()).max())
check("quantum_circuit_simulation(n12, 40 bitstrings) == reference (same planner settings; oracle as executor)",
      err < 1e-5 and list(gbits) == list(wbits), rel_err=err, seconds=round(time.time() - t0, 1))
want, _ = R.quantum_circuit_simulation(N12, [], 30, trial_num=2)
got, _ = A.quantum_circuit_simulation(N12, [], 30, trial_num=2, device="cpu")
err = float(np.abs(got.reshape(-1).numpy() - want.reshape(-1).numpy()).max() / np.abs(want.numpy()).max())
check("quantum_circuit_simulation(n12, full amplitude) == reference", err < 1e-5, rel_err=err)

record["all_passed"] = all(c["passed"] for c in record["checks"])
# the sources this record vouches for (tests/test_boundary_record.py refuses a record older than they are)
record["source_sha16"] = S.boundary_source_sha16()
with open(os.path.join(HERE, "boundary_check.json"), "w") as f:
    json.dump(record, f, indent=1)
print(f"{len(record['checks'])} boundary checks passed -> tests/golden/boundary_check.json")
