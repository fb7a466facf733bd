#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONHASHSEED=0 PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [case ...]

Every fixture is data: leaf tensors, the compiled scheme (the step tuples produced by
/root/reference/artensor/contraction.py:23-59 and :208-341), slicing indices
(/root/reference/artensor/simulation.py:60-65) and the outputs the reference's own
executors (/root/reference/artensor/contraction.py:62-76, :132-205, the slice loop
simulation.py:107-114) returned here on torch-CPU.  No reference source is stored.

PYTHONHASHSEED must be 0: bond labels are strings that pass through set()
(contraction.py:17,:49), so schemes depend on the hash seed.
"""
import os
import sys
import time
import types
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

import artensor  # the reference  # noqa: E402
from artensor import (  # noqa: E402
    TensorNetworkCircuit, TensorNetworkSimulation, AbstractTensorNetwork, ContractionTree,
    find_order, contraction_scheme, tensor_contraction, contraction_scheme_sparse,
    tensor_contraction_sparse,
)
from artensor_amd.fixtures import save_case  # noqa: E402

PLAN = dict(trials=5, iters=10, slicing_repeat=1)
N12_QSIM = os.path.join(REF, "tests", "circuit_n12_m14_s0_e0_pEFGH.qsim")
N30_PY = os.path.join(REF, "examples", "circuit_n30_m14_s0_e0_pEFGH.py")
N30_AMPS = os.path.join(REF, "examples", "amplitudes_n30_m14_s0_e0_pEFGH_10000.txt")
N30_QSIM = "/tmp/circuit_n30_m14_s0_e0_pEFGH.qsim"

# known-answer table of the reference's own test (tests/test_circuits.py:25-31)
N12_TABLE = {
    "100001000001": 0.0198028199 + 1j * (0.0106442748),
    "000101111011": 0.00497586094 + 1j * (-0.0245072283),
    "011000101100": -0.00853562169 + 1j * (-0.00701293815),
    "111001100001": -0.0100137182 + 1j * (0.0147468708),
    "001110110000": 0.00681955926 + 1j * (0.0106616206),
}


# --------------------------------------------------------------------------------------
# n30 circuit: the reference only ships it as a cirq program; cirq is absent here.  The
# program uses five gate constructors only, each of which maps 1:1 onto a qsim text gate
# understood by artensor/circuit.py:5-15.  Evaluate the program against a recording
# stand-in and emit qsim text to /tmp (an input-format conversion; nothing is stored).
# --------------------------------------------------------------------------------------
def n30_qsim():
    if os.path.exists(N30_QSIM):
        return N30_QSIM

    class Q:
        def __init__(s, r, c): s.rc = (r, c)
        def __eq__(s, o): return s.rc == o.rc
        def __hash__(s): return hash(s.rc)

    class Op:
        def __init__(s, name, params=()): s.name, s.params = name, params
        def on(s, *qs): return (s.name, qs, s.params)

    class Pow:
        def __init__(s, name): s.name = name
        def __pow__(s, e):
            assert e == 0.5
            return Op(s.name)

    def phasedx(phase_exponent, exponent):
        assert (phase_exponent, exponent) == (0.25, 0.5)
        return Op("hz_1_2")

    cirq = types.ModuleType("cirq")
    cirq.GridQubit = Q
    cirq.X, cirq.Y = Pow("x_1_2"), Pow("y_1_2")
    cirq.PhasedXPowGate = phasedx
    cirq.Rz = lambda rads: Op("rz", (rads,))
    cirq.FSimGate = lambda theta, phi: Op("fs", (theta, phi))
    cirq.Moment = lambda operations: list(operations)
    cirq.Circuit = lambda moments: list(moments)
    sys.modules["cirq"] = cirq
    ns = {}
    exec(compile(open(N30_PY).read(), N30_PY, "exec"), ns)
    del sys.modules["cirq"]
    order = {q.rc: i for i, q in enumerate(ns["QUBIT_ORDER"])}
    lines = [str(len(order))]
    for layer, moment in enumerate(ns["CIRCUIT"]):
        for name, qs, params in moment:
            lines.append(" ".join([str(layer), name] + [str(order[q.rc]) for q in qs]
                                  + [repr(float(p)) for p in params]))
    with open(N30_QSIM, "w") as f:
        f.write("\n".join(lines) + "\n")
    return N30_QSIM


def read_google(n):
    bits, amps = [], []
    with open(N30_AMPS) as f:
        for line in f:
            b, re, im = line.split()
            bits.append(b)
            amps.append(float(re) + 1j * float(im))
            if len(bits) == n:
                break
    return bits, np.array(amps)


def plan(qsim, bitstrings, sc_target):
    sim = TensorNetworkSimulation.from_circuit_file(qsim, bitstrings)
    t0 = time.time()
    sim.prepare_contraction(sc_target=sc_target, **PLAN)
    tc, sc, mc = sim.ctree.tree_complexity()[:3]
    meta = dict(sc_target=sc_target, plan=PLAN, log10_tc=float(tc), sc=float(sc),
                plan_seconds=time.time() - t0, output_bonds=[str(b) for b in sim.output_bonds],
                pattern=sim.pattern, hashseed=0, torch=torch.__version__)
    if hasattr(sim, "permute_dims"):
        meta["permute_dims"] = [int(x) for x in sim.permute_dims]
    if sim.pattern == "sparse":
        meta["bitstrings_sorted"] = list(sim.bitstrings_sorted)
    return sim, meta


def slicing_ok(sim):
    """True when the reference slice loop is well defined (a tensor
    carrying two sliced bonds makes the second select() use a stale dim index)."""
    seen = {}
    for bond, lst in sim.slicing_indices.items():
        for tid, ind in lst:
            seen.setdefault(tid, []).append(ind)
    for tid, inds in seen.items():
        for a in range(len(inds)):
            for b in range(a + 1, len(inds)):
                if inds[a] < inds[b]:
                    return False
    return True


# --------------------------------------------------------------------------------------
def case_n12_dense():
    sim, meta = plan(N12_QSIM, [], 30)
    assert len(sim.slicing_indices) == 0
    tensors = {i: sim.tensors[i].to(torch.complex64) for i in sim.tensors}
    raw = tensor_contraction(dict(tensors), sim.scheme)
    final = sim.contraction().reshape(-1)
    assert torch.equal(raw.permute(sim.permute_dims).reshape(-1), final)
    sv = TensorNetworkCircuit(N12_QSIM).state_vec().reshape(-1)
    meta["table"] = {b: [v.real, v.imag] for b, v in N12_TABLE.items()}
    save_case(os.path.join(HERE, "n12_dense.npz"), tensors, sim.scheme, meta,
              arrays=dict(raw=raw, final=final, state_vec=sv))
    print("n12_dense", len(sim.scheme), "steps; max|tn-sv| =", (final - sv).abs().max().item())


def case_n12_sparse5():
    bits = list(N12_TABLE.keys())
    sim, meta = plan(N12_QSIM, bits, 30)
    tensors = {i: sim.tensors[i].to(torch.complex64) for i in sim.tensors}
    out = sim.contraction()
    meta["table"] = {b: [v.real, v.imag] for b, v in N12_TABLE.items()}
    save_case(os.path.join(HERE, "n12_sparse5.npz"), tensors, sim.scheme, meta,
              arrays=dict(final=out), slicing_indices=sim.slicing_indices)
    ref = np.array([N12_TABLE[b] for b in sim.bitstrings_sorted])
    print("n12_sparse5 rel err vs table", np.abs(out.numpy() - ref).max() / np.abs(ref).max())


def case_n12_sparse_sliced():
    rng = np.random.RandomState(7)
    bits = sorted({np.binary_repr(x, 12) for x in rng.randint(0, 4096, size=24)})[:20]
    for sc_target in (10, 9, 11, 8):
        sim, meta = plan(N12_QSIM, bits, sc_target)
        if len(sim.slicing_indices) >= 2 and slicing_ok(sim):
            break
    else:
        raise SystemExit("no well-defined sliced n12 sparse case found")
    tensors = {i: sim.tensors[i].to(torch.complex64) for i in sim.tensors}
    out = sim.contraction()
    sv = TensorNetworkCircuit(N12_QSIM).state_vec().reshape(-1)
    want = np.array([sv[int(b, 2)].item() for b in sim.bitstrings_sorted])
    # per-slice outputs too, so the slice loop can be checked term by term
    save_case(os.path.join(HERE, "n12_sparse_sliced.npz"), tensors, sim.scheme, meta,
              arrays=dict(final=out, state_vec_at=want), slicing_indices=sim.slicing_indices)
    print("n12_sparse_sliced", len(sim.slicing_indices), "bonds, sc", sc_target,
          "err vs sv", np.abs(out.numpy() - want).max())


def case_n12_scinot():
    """scientific_notation=True branch (contraction.py:197-204)."""
    bits = list(N12_TABLE.keys())
    sim, meta = plan(N12_QSIM, bits, 30)
    tensors = {i: sim.tensors[i].to(torch.complex64) for i in sim.tensors}
    factor, out = tensor_contraction_sparse(dict(tensors), sim.scheme, scientific_notation=True)
    save_case(os.path.join(HERE, "n12_sparse5_scinot.npz"), tensors, sim.scheme, meta,
              arrays=dict(final=out, factor=factor))
    print("n12_scinot factor", factor)


def _rand_tn(nv, D, seed, n_open):
    """3-regular random graph TN (SURVEY 8d input 2).  Bonds are ints; n_open extra
    dangling bonds on the first tensors make it an open network."""
    import networkx as nx
    g = nx.random_regular_graph(3, nv, seed=seed)
    edges = sorted((min(a, b), max(a, b)) for a, b in g.edges())
    tensor_bonds = {i: [] for i in range(nv)}
    for e, (a, b) in enumerate(edges):
        tensor_bonds[a].append(e)
        tensor_bonds[b].append(e)
    nb = len(edges)
    for k in range(n_open):
        tensor_bonds[k].append(nb + k)
    bond_dims = {b: float(D) for b in range(nb + n_open)}
    gen = torch.Generator().manual_seed(seed)
    tensors = {}
    for i in range(nv):
        shape = [D] * len(tensor_bonds[i])
        tensors[i] = torch.complex(torch.randn(shape, generator=gen),
                                   torch.randn(shape, generator=gen)) / D ** 1.5
    return tensors, tensor_bonds, bond_dims


def case_random_tns():
    try:
        import networkx  # noqa: F401
    except ImportError:
        print("networkx missing: random TN cases skipped")
        return
    for nv, D, n_open, sc_target, name in [
        (16, 2, 0, 30, "rand_D2_closed"),
        (14, 3, 2, 30, "rand_D3_open"),
        (12, 4, 0, 30, "rand_D4_closed"),
        (16, 2, 3, 4, "rand_D2_open_sliced"),
        (24, 2, 0, 3, "rand_D2_closed_sliced"),
    ]:
        tensors, tensor_bonds, bond_dims = _rand_tn(nv, D, 0, n_open)
        order, slicing_bonds, ctree = find_order(
            deepcopy(tensor_bonds), deepcopy(bond_dims), [], 0, 1, sc_target=sc_target,
            trials=4, iters=5, betas=np.linspace(3.0, 21.0, 61), start_seed=0, slicing_repeat=1)
        scheme, output_bonds = contraction_scheme(deepcopy(ctree))
        # slice loop restated as simulation.py:198-213 does it
        slicing_indices = {}
        for bond in slicing_bonds:
            slicing_indices[bond] = [(tid, tensor_bonds[tid].index(bond))
                                     for tid in tensor_bonds if bond in tensor_bonds[tid]]
        ok = True
        seen = {}
        for bond, lst in slicing_indices.items():
            for tid, ind in lst:
                if any(prev < ind for prev in seen.get(tid, [])):
                    ok = False
                seen.setdefault(tid, []).append(ind)
        if not ok:
            print(name, "skipped: reference slice loop ill-defined for this plan")
            continue
        shape = [int(bond_dims[b]) for b in output_bonds]
        collect = torch.zeros(shape, dtype=torch.complex64)
        for s in range(2 ** len(slicing_bonds)):
            cfg = list(map(int, np.binary_repr(s, len(slicing_bonds)))) if slicing_bonds else []
            sliced = dict(tensors)
            for x, bond in enumerate(slicing_bonds):
                for tid, ind in slicing_indices[bond]:
                    sliced[tid] = sliced[tid].select(ind, cfg[x]).clone()
            collect += tensor_contraction(sliced, scheme)
        # independent one-shot complex128 einsum over the whole network
        labels = sorted(bond_dims.keys())
        sub = []
        for i in range(nv):
            sub += [tensors[i].to(torch.complex128), list(tensor_bonds[i])]
        exact = torch.einsum(*sub, list(output_bonds))
        err = (collect.to(torch.complex128) - exact).abs().max() / exact.abs().max()
        tc = ctree.tree_complexity()[0]
        meta = dict(D=D, nv=nv, n_open=n_open, sc_target=sc_target, log10_tc=float(tc),
                    output_bonds=[int(b) for b in output_bonds], bond_dim=D,
                    n_slicing=len(slicing_bonds))
        save_case(os.path.join(HERE, name + ".npz"), tensors, scheme, meta,
                  arrays=dict(final=collect, exact128=exact), slicing_indices=slicing_indices)
        print(name, "steps", len(scheme), "slices", 2 ** len(slicing_bonds), "rel err vs c128", err.item())


def case_n30_plan():
    """n30 m14 full amplitude: leaf tensors + 180-step scheme only (fast)."""
    sim, meta = plan(n30_qsim(), [], 30)
    assert len(sim.slicing_indices) == 0
    tensors = {i: sim.tensors[i].to(torch.complex64) for i in sim.tensors}
    bits, amps = read_google(10000)
    meta["google_bitstrings"] = bits
    save_case(os.path.join(HERE, "n30_dense.npz"), tensors, sim.scheme, meta,
              arrays=dict(google=amps))
    print("n30_dense plan", len(sim.scheme), "steps log10 tc", meta["log10_tc"])
    return sim, meta


def case_n30_run():
    """Run the reference dense executor on n30 (minutes, ~25 GB) and append statistics."""
    from artensor_amd.fixtures import load_case
    path = os.path.join(HERE, "n30_dense.npz")
    case = load_case(path)
    tensors = case.fresh_tensors()
    t0 = time.time()
    raw = tensor_contraction(tensors, case.scheme)
    wall = time.time() - t0
    final = raw.permute(case.meta["permute_dims"]).reshape(-1)
    bits = case.meta["google_bitstrings"]
    pos = torch.tensor([int(b, 2) for b in bits])
    at = final[pos].clone()
    blocks = final.reshape(1024, -1).sum(dim=1).to(torch.complex128)
    norm2 = float((final.real.double() ** 2 + final.imag.double() ** 2).sum())
    stride = final[:: 2 ** 14 + 1][:65536].clone()   # 65536-point strided probe
    meta = case.meta
    meta["reference_cpu_seconds"] = wall
    meta["reference_cpu_threads"] = torch.get_num_threads()
    meta["norm2"] = norm2
    for k in ("tensor_ids", "steps"):
        meta.pop(k, None)
    arrays = dict(case.arrays)
    arrays.update(amps_at_google=at, block_sums=blocks, strided=stride)
    save_case(path, case.tensors, case.scheme, meta, arrays=arrays)
    g = case.arrays["google"]
    print("n30 run", wall, "s; norm2", norm2, "max rel err vs google",
          (np.abs(at.numpy() - g) / np.abs(g)).max())


def case_n30_sparse(nbits):
    bits, amps = read_google(nbits)
    sim, meta = plan(n30_qsim(), bits, 30)
    assert len(sim.slicing_indices) == 0
    tensors = {i: sim.tensors[i].to(torch.complex64) for i in sim.tensors}
    t0 = time.time()
    out = sim.contraction()
    meta["reference_cpu_seconds"] = time.time() - t0
    g = np.array([amps[bits.index(b)] for b in sim.bitstrings_sorted])
    save_case(os.path.join(HERE, f"n30_sparse{nbits}.npz"), tensors, sim.scheme, meta,
              arrays=dict(final=out, google=g))
    print("n30_sparse", nbits, meta["reference_cpu_seconds"], "s  max rel err vs google",
          (np.abs(out.numpy() - g) / np.abs(g)).max())


def case_n30_sliced(k=3):
    """n30 full amplitude with k manually sliced inner bonds, the way the reference's
    notebook does it (examples/sycamore.ipynb cell 7: AbstractTensorNetwork.slicing +
    ContractionTree(tn, order) + contraction_scheme).  Plan + leaf tensors only; the sum of
    the 2**k slices equals the unsliced n30 result, which n30_dense.npz pins."""
    from artensor.simulation import get_bond_tensors
    sim = TensorNetworkSimulation.from_circuit_file(n30_qsim(), [])
    betas = np.linspace(3.0, 21.0, 61)
    order, slicing_bonds, ctree = find_order(
        sim.tensor_bonds, sim.bond_dims, sim.final_qubits, 0, 1, sc_target=30,
        betas=betas, start_seed=0, alpha=32.0, **PLAN)
    assert not slicing_bonds
    bond_tensors = get_bond_tensors(sim.tensor_bonds)
    scheme0, out0 = contraction_scheme(deepcopy(ctree))
    # choose inner bonds (shared by two tensors, not output bonds) that live longest in
    # the big "state" tensor: here simply the k bonds contracted by the last big steps
    # that are leaf-leaf bonds, greedily keeping the slice loop well defined.
    tn = AbstractTensorNetwork(deepcopy(sim.tensor_bonds), deepcopy(sim.bond_dims),
                               sim.final_qubits, 1)
    cand = [b for b, ts in bond_tensors.items() if len(ts) == 2 and b not in out0]
    # rank candidates by how much slicing them lowers tc of the tree
    scored = []
    for b in cand:
        t2 = deepcopy(tn)
        t2.slicing(b)
        ct = ContractionTree(t2, order, 0)
        scored.append((ct.tree_complexity()[0], b))
    scored.sort()
    chosen, used = [], {}
    for _, b in scored:
        inds = [(tid, sim.tensor_bonds[tid].index(b)) for tid in bond_tensors[b]]
        if any(tid in used for tid, _ in inds):
            continue
        chosen.append(b)
        for tid, _ in inds:
            used[tid] = True
        if len(chosen) == k:
            break
    for b in chosen:
        tn.slicing(b)
    ctree_s = ContractionTree(tn, order, 0)
    scheme, output_bonds = contraction_scheme(deepcopy(ctree_s))
    assert list(output_bonds) == list(out0) or sorted(output_bonds) == sorted(out0)
    slicing_indices = {b: [(tid, sim.tensor_bonds[tid].index(b)) for tid in bond_tensors[b]]
                       for b in chosen}
    bond_inds = []
    bt = get_bond_tensors(sim.tensor_bonds)
    for x in range(len(output_bonds)):
        tid = list(bt[output_bonds[x]])[0]
        bond_inds.append(list(sim.final_qubits).index(tid))
    tc = ctree_s.tree_complexity()[0]
    meta = dict(sc_target=30, plan=PLAN, log10_tc_per_slice=float(tc), n_slicing=k,
                permute_dims=[int(x) for x in np.argsort(bond_inds)],
                output_bonds=[str(b) for b in output_bonds], hashseed=0)
    tensors = {i: sim.tensors[i].to(torch.complex64) for i in sim.tensors}
    save_case(os.path.join(HERE, f"n30_dense_sliced{k}.npz"), tensors, scheme, meta,
              slicing_indices=slicing_indices)
    print("n30_sliced", k, "bonds", chosen, "log10 tc/slice", tc)


def case_n30_dense_parts():
    """The unsliced n30 m14 contraction over N = 2, 4, 8 ranks WITHOUT a collective: rank r computes the slab of 2^30 / N
    amplitudes whose log2 N chosen output qubits spell r.  Round 4 fixed those qubits at the leaves of the ONE tree
    planned for the full network (artensor_amd.partition_output: 1.36 / 2.13 / 3.78 x the unsliced FLOP at N = 2 / 4 /
    8); here the REDUCED network -- those output bonds removed from their leaves -- goes back to the reference's planner
    (order_finder.py:174-198, same budget as the full plan), one plan per N (the slabs of one N differ in leaf VALUES
    only).  Several choices of qubits are planned and the cheapest kept.  Expected values: the full fixture's amplitudes
    at Google's bitstrings, which the reference computed (n30_dense.npz)."""
    from artensor_amd.fixtures import load_case
    full = load_case(os.path.join(HERE, "n30_dense.npz"))
    tc_full = full.meta["log10_tc"]
    sim0 = TensorNetworkSimulation.from_circuit_file(n30_qsim(), [])
    fq = sorted(sim0.final_qubits) if isinstance(sim0.final_qubits, (set, frozenset)) else list(sim0.final_qubits)
    assert len(fq) == 30
    count = {}
    for tid, bonds in sim0.tensor_bonds.items():
        for b in bonds:
            count[b] = count.get(b, 0) + 1
    dangling = {}
    for q, tid in enumerate(fq):
        mine = [b for b in sim0.tensor_bonds[tid] if count[b] == 1]
        assert len(mine) == 1, (tid, mine)
        dangling[q] = mine[0]
    for k in (1, 2, 3):
        N = 2 ** k
        # every run of k neighbouring qubits (in the reference's final-qubit order) + one spread choice: the planner's result
        # depends strongly on WHICH qubits are fixed (N = 8: 0.97 x for qubits 27-29, 11 x for 14-16), and a plan is 12 s
        cands = [tuple(range(q, q + k)) for q in range(0, 31 - k)] + [tuple(int(round(x)) for x in np.linspace(0, 29, k + 2)[1:-1])]
        best = None
        for cand in dict.fromkeys(cands):
            tb = {t: list(b) for t, b in sim0.tensor_bonds.items()}
            fixed = []
            for q in cand:
                tid = fq[q]
                dim = tb[tid].index(dangling[q])
                fixed.append((int(tid), int(dim), int(q)))
            for tid, dim, q in sorted(fixed, key=lambda x: -x[1]):
                tb[tid].pop(dim)
            fq2 = [t for q, t in enumerate(fq) if q not in cand]
            bd = {b: 2.0 for b in set().union(*tb.values())}
            t0 = time.time()
            order, slicing_bonds, ctree = find_order(
                deepcopy(tb), deepcopy(bd), set(fq2), 0, 1, sc_target=30, betas=np.linspace(3.0, 21.0, 61), start_seed=0, **PLAN)
            tc, sc = ctree.tree_complexity()[:2]
            print(f"N={N} fixed qubits {cand}: log10 tc {tc:.3f} sc {sc:.1f} sliced {len(slicing_bonds)} overhead "
                  f"{N * 10 ** (tc - tc_full):.2f} x  ({time.time() - t0:.0f} s)", flush=True)
            if len(slicing_bonds) == 0 and (best is None or tc < best[0]):
                best = (float(tc), float(sc), cand, fixed, fq2, ctree)
        tc, sc, cand, fixed, fq2, ctree = best
        scheme, output_bonds = contraction_scheme(deepcopy(ctree))
        # raw result dim x <-> the qubit (0 = first final qubit, most significant in the reference's final order)
        out_qubits = []
        for b in output_bonds:
            q = [q for q in range(30) if dangling[q] == b]
            assert len(q) == 1
            out_qubits.append(int(q[0]))
        tensors = {i: sim0.tensors[i].to(torch.complex64) for i in sim0.tensors}
        meta = dict(n_slabs=N, fixed=[list(x) for x in fixed], fixed_qubits=[int(q) for q in cand], out_qubits=out_qubits,
                    log10_tc=tc, sc=sc, log10_tc_full=float(tc_full), executed_flop_over_unsliced=float(N * 10 ** (tc - tc_full)),
                    plan=PLAN, hashseed=0,
                    note="slab r: fixed[j] = (leaf tensor id, dim, qubit) takes bit j of r (bit 0 = first entry); the scheme runs on "
                         "the leaves after those selects; raw result dim x holds qubit out_qubits[x]")
        save_case(os.path.join(HERE, f"n30_dense_part{N}.npz"), tensors, scheme, meta)
        print(f"n30_dense_part{N}: qubits {cand}, {len(scheme)} steps, overhead {meta['executed_flop_over_unsliced']:.2f} x", flush=True)


def _dump_tree(ctree):
    """Planner products as plain data (what contraction.py:23-59 / :208-341 consume):
    vertices with their bond lists IN THE ITERATION ORDER the reference saw."""
    ctree.mark_rep_tensor()
    verts = {}

    def walk(v):
        key = ",".join(map(str, sorted(v.contain_tensors)))
        verts[key] = dict(contain_tensors=sorted(int(x) for x in v.contain_tensors),
                          contain_bonds=[str(b) for b in list(v.contain_bonds)], sc=float(v.sc),
                          left=None, right=None)
        if v.left and v.right:
            verts[key]["left"] = walk(v.left)
            verts[key]["right"] = walk(v.right)
        return key

    root = walk(ctree.tree[ctree.all_tensors])
    fq = ctree.tn.final_qubits
    return dict(root=root, vertices=verts,
                tensor_bonds={str(k): [str(b) for b in v] for k, v in ctree.tn.tensor_bonds.items()},
                final_qubits=sorted(int(x) for x in fq) if fq is not None else [])


def case_trees():
    """Trees + the schemes the reference compiled from them (scheme-compiler parity)."""
    import json
    out = {}
    sim, _ = plan(N12_QSIM, [], 30)
    scheme, output_bonds = contraction_scheme(deepcopy(sim.ctree))
    out["n12_dense"] = dict(tree=_dump_tree(deepcopy(sim.ctree)), output_bonds=[str(b) for b in output_bonds],
                            scheme=[[list(map(int, e)), eq] for e, eq in scheme])
    rng = np.random.RandomState(3)
    bits = sorted({np.binary_repr(x, 12) for x in rng.randint(0, 4096, size=40)})
    for name, sc_target in (("n12_sparse", 30), ("n12_sparse_chunked", 8), ("n12_sparse_chunked6", 6)):
        sim, _ = plan(N12_QSIM, bits, 30)
        scheme, bonds, sorted_bits = contraction_scheme_sparse(deepcopy(sim.ctree), bits, sc_target=sc_target)
        steps = []
        for st in scheme:
            rec = dict(edge=list(map(int, st[0])), eq=st[1],
                       batch=[[np.asarray(x).tolist() for x in st[2][0]], [np.asarray(x).tolist() for x in st[2][1]]])
            if len(st) == 5:
                rec["rshape"] = None if st[3] is None else list(st[3])
                rec["next_shape"] = list(st[4])
            steps.append(rec)
        out[name] = dict(tree=_dump_tree(deepcopy(sim.ctree)), bitstrings=bits, sc_target=sc_target,
                         bonds=[str(b) for b in bonds], bitstrings_sorted=list(sorted_bits), scheme=steps)
        print(name, len(scheme), "steps,", sum(len(s) == 5 for s in scheme), "5-tuples,",
              sum(len(s[2][0]) > 1 for s in scheme), "chunked")
    with open(os.path.join(HERE, "trees.json"), "w") as f:
        json.dump(out, f)


N53_M20 = os.path.join(REF, "examples", "circuits", "circuit_n53_m20_s0_e0_pABCDCDAB.qsim")
N53_M14 = "/tmp/circuit_n53_m14_trunc.qsim"


def n53_m14_qsim():
    """BASELINE config 4 names Sycamore n53 m14; the reference bundles only m20.  Derivation
    (SURVEY.md 8c): the first 14 cycles (layers 0..55) of the bundled m20 file plus its final
    single-qubit layer (80) renumbered 56.  Input-format conversion to /tmp, nothing stored."""
    if not os.path.exists(N53_M14):
        out = []
        with open(N53_M20) as f:
            lines = f.read().splitlines()
        out.append(lines[0])
        for ln in lines[1:]:
            parts = ln.split()
            layer = int(parts[0])
            if layer < 56:
                out.append(ln)
            elif layer == 80:
                out.append(" ".join(["56"] + parts[1:]))
        with open(N53_M14, "w") as f:
            f.write("\n".join(out) + "\n")
    return N53_M14


def case_n53_plan(which="m14"):
    """n53 m14 (derived) or the bundled n53 m20, one bitstring, sc_target 30: plan + leaf tensors +
    slicing indices."""
    bits = ["0" * 53]
    qsim = n53_m14_qsim() if which == "m14" else os.path.join(REF, "examples", "circuits", "circuit_n53_m20_s0_e0_pABCDCDAB.qsim")
    sim, meta = plan(qsim, bits, 30)
    # With a single bitstring the chunking rule of contraction.py:288-297 (chunks when
    # log2(rows) + rank > sc_target - 2) produces chunks of int(1 / 8) = 0 rows and the
    # reference executor then dies on an empty tensor.  Recompile the scheme from the same
    # tree with the chunk threshold out of reach; the tree / slicing stay those of sc_target 30.
    sim.update_scheme(40, bits)
    assert all(len(st[2][0]) <= 1 for st in sim.scheme)
    meta["scheme_chunk_threshold"] = 40
    meta["n_slicing"] = len(sim.slicing_indices)
    meta["reference_slice_loop_well_defined"] = bool(slicing_ok(sim))
    meta["derivation"] = ("first 14 cycles of circuit_n53_m20_s0_e0_pABCDCDAB.qsim + its final 1-qubit layer" if which == "m14"
                          else "circuit_n53_m20_s0_e0_pABCDCDAB.qsim as bundled")
    tensors = {i: sim.tensors[i].to(torch.complex64) for i in sim.tensors}
    # The reference computes a sliced bond's dim as its position in tensor_bonds[tid]
    # (simulation.py:62-65), but in the sparse pattern a final-qubit tensor carries a leading
    # batch dim that tensor_bonds does not list (shape (2, 2) for bonds ['35-0']): select()
    # then hits the batch dim.  Stored here are the dims of the actual tensors.
    fixed, touched = {}, 0
    for bond, lst in sim.slicing_indices.items():
        fixed[bond] = []
        for tid, ind in lst:
            shift = tensors[tid].dim() - len(sim.tensor_bonds[tid])
            touched += shift != 0
            fixed[bond].append((tid, ind + shift))
    meta["slicing_dims_shifted_for_batch_dim"] = int(touched)
    save_case(os.path.join(HERE, f"n53_{which}_sliced.npz"), tensors, sim.scheme, meta, slicing_indices=fixed)
    print(f"n53_{which} plan:", len(sim.scheme), "steps,", len(sim.slicing_indices), "sliced bonds, log10 tc/slice",
          meta["log10_tc"], "reference loop well defined:", meta["reference_slice_loop_well_defined"])


def case_n53_slice0(which="m14"):
    """Reference sparse executor on slice 0 of the n53 plan (CPU, ~half an hour, ~25 GB).
    The slice is applied with the unsliced-index semantics (artensor_amd.apply_slice), which is
    what the reference loop means wherever it is well defined."""
    from artensor_amd.fixtures import load_case
    from artensor_amd.simulation import apply_slice, slice_assignments
    path = os.path.join(HERE, f"n53_{which}_sliced.npz")
    case = load_case(path)
    out = {}
    for s in (0,):
        cfg = slice_assignments(len(case.slicing_indices), s)
        sliced = apply_slice(case.fresh_tensors(), case.slicing_indices, cfg)
        t0 = time.time()
        res = tensor_contraction_sparse(sliced, case.scheme)
        out[f"slice{s}"] = res.reshape(-1).numpy().copy()
        print("n53 slice", s, res.reshape(-1), time.time() - t0, "s", flush=True)
    meta = case.meta
    for k in ("tensor_ids", "steps", "slicing_indices"):
        meta.pop(k, None)
    meta["reference_cpu_seconds_per_slice"] = time.time() - t0
    save_case(path, case.tensors, case.scheme, meta, arrays=out, slicing_indices=case.slicing_indices)


def n53_batch_bitstrings(n_open=16, count=1024, seed=53):
    """BASELINE configs[4] ("big-batch sampling"): `count` distinct bitstrings that agree on 53 - n_open
    closed qubits (all '0') and vary on n_open randomly placed open ones -- a strict subset of the
    2^n_open product, so the sparse compiler has to emit row selects / gathered steps."""
    rng = np.random.RandomState(seed)
    open_pos = np.sort(rng.choice(53, size=n_open, replace=False))
    seen, bits = set(), []
    while len(bits) < count:
        x = int(rng.randint(0, 2 ** n_open))
        if x in seen:
            continue
        seen.add(x)
        s = ["0"] * 53
        for k, p in enumerate(open_pos):
            s[p] = str((x >> k) & 1)
        bits.append("".join(s))
    return bits, [int(p) for p in open_pos]


def case_n53m20_batch_plan(sc_target=30):
    """The bundled n53 m20 circuit with a batch of 1 024 correlated bitstrings (configs[4]): plan,
    leaf tensors, sparse scheme with its index tensors, slicing indices."""
    bits, open_pos = n53_batch_bitstrings()
    sim, meta = plan(N53_M20, bits, sc_target)
    kinds = dict(A=0, B=0, C=0, C_select=0, D=0)
    for st in sim.scheme:
        if len(st[2][0]) > 1:
            kinds["A"] += 1
        elif len(st) > 3 and len(st[2][0]) == len(st[2][1]) == 1:
            kinds["B"] += 1
        elif len(st) > 3:
            kinds["C"] += 1
            kinds["C_select"] += len(st[2][0]) == 1
        else:
            kinds["D"] += 1
    meta["branches"] = kinds
    meta["open_qubits"] = open_pos
    meta["n_slicing"] = len(sim.slicing_indices)
    meta["reference_slice_loop_well_defined"] = bool(slicing_ok(sim))
    meta["derivation"] = "circuit_n53_m20_s0_e0_pABCDCDAB.qsim as bundled; 1024 bitstrings over 16 open qubits"
    tensors = {i: sim.tensors[i].to(torch.complex64) for i in sim.tensors}
    fixed, touched = {}, 0   # see case_n53_plan: dims of the actual (batch-carrying) tensors
    for bond, lst in sim.slicing_indices.items():
        fixed[bond] = []
        for tid, ind in lst:
            shift = tensors[tid].dim() - len(sim.tensor_bonds[tid])
            touched += shift != 0
            fixed[bond].append((tid, ind + shift))
    meta["slicing_dims_shifted_for_batch_dim"] = int(touched)
    save_case(os.path.join(HERE, "n53_m20_batch.npz"), tensors, sim.scheme, meta, slicing_indices=fixed)
    print("n53_m20_batch plan:", len(sim.scheme), "steps,", len(sim.slicing_indices), "sliced bonds, log10 tc/slice",
          meta["log10_tc"], "sc", meta["sc"], "branches", kinds, flush=True)


def case_n53m20_batch_slice0():
    """Reference sparse executor (contraction.py:132-205) on slice 0 of the big-batch plan, torch-CPU."""
    from artensor_amd.fixtures import load_case
    from artensor_amd.simulation import apply_slice, slice_assignments
    path = os.path.join(HERE, "n53_m20_batch.npz")
    case = load_case(path)
    cfg = slice_assignments(len(case.slicing_indices), 0)
    sliced = apply_slice(case.fresh_tensors(), case.slicing_indices, cfg)
    t0 = time.time()
    res = tensor_contraction_sparse(sliced, case.scheme)
    dt = time.time() - t0
    print("n53 m20 batch slice 0:", tuple(res.shape), f"{dt:.0f} s", flush=True)
    meta = case.meta
    for k in ("tensor_ids", "steps", "slicing_indices"):
        meta.pop(k, None)
    meta["reference_cpu_seconds_per_slice"] = dt
    meta["reference_cpu_threads"] = torch.get_num_threads()
    save_case(path, case.tensors, case.scheme, meta, arrays=dict(slice0=res.reshape(-1).numpy().copy()),
              slicing_indices=case.slicing_indices)


def case_n53m20_bigbatch(n_open=17, count=65536):
    """configs[4] at a batch that deserves the name (round 5): the bundled n53 m20 circuit with 2^16 correlated
    bitstrings -- a random HALF of the 2^17 product over 17 open qubits, so the sparse compiler emits row selects and
    gathered steps with index lists of tens of thousands of rows.  The reference's own compiler compares substrings
    bitstring by bitstring (contraction.py:249-283: quadratic in the batch, hours here), so the scheme is compiled by
    artensor_amd.contraction_scheme_sparse -- its vectorised restatement, byte-identical to the reference's output on
    every branch for the batches the reference can compile (tests/golden/check_boundary.py, 1 500 bitstrings) --
    from the REFERENCE's tree; planning and the executor that produces the expected slice are the reference's."""
    import artensor.simulation as RS
    import artensor_amd
    from artensor_amd.simulation import apply_slice, slice_assignments
    bits, open_pos = n53_batch_bitstrings(n_open=n_open, count=count, seed=53)
    ref_compiler = RS.contraction_scheme_sparse
    RS.contraction_scheme_sparse = artensor_amd.contraction_scheme_sparse
    try:
        t0 = time.time()
        sim, meta = plan(N53_M20, bits, 30)
        meta["plan_and_compile_seconds"] = time.time() - t0
    finally:
        RS.contraction_scheme_sparse = ref_compiler
    kinds = dict(A=0, B=0, C=0, C_select=0, D=0)
    for st in sim.scheme:
        if len(st[2][0]) > 1:
            kinds["A"] += 1
        elif len(st) > 3 and len(st[2][0]) == len(st[2][1]) == 1:
            kinds["B"] += 1
        elif len(st) > 3:
            kinds["C"] += 1
            kinds["C_select"] += len(st[2][0]) == 1
        else:
            kinds["D"] += 1
    meta["branches"] = kinds
    meta["open_qubits"] = open_pos
    meta["n_slicing"] = len(sim.slicing_indices)
    meta["scheme_compiled_by"] = "artensor_amd.contraction_scheme_sparse (vectorised restatement) on the reference's tree"
    meta["derivation"] = f"circuit_n53_m20_s0_e0_pABCDCDAB.qsim as bundled; {count} bitstrings over {n_open} open qubits"
    tensors = {i: sim.tensors[i].to(torch.complex64) for i in sim.tensors}
    fixed = {}
    for bond, lst in sim.slicing_indices.items():
        fixed[bond] = [(tid, ind + tensors[tid].dim() - len(sim.tensor_bonds[tid])) for tid, ind in lst]
    print("n53_m20_bigbatch plan:", len(sim.scheme), "steps,", len(sim.slicing_indices), "sliced bonds, log10 tc/slice",
          meta["log10_tc"], "sc", meta["sc"], "branches", kinds, f"{meta['plan_and_compile_seconds']:.0f} s", flush=True)
    cfg = slice_assignments(len(fixed), 0)
    sliced = apply_slice({i: t.clone() for i, t in tensors.items()}, fixed, cfg)
    t0 = time.time()
    res = tensor_contraction_sparse(sliced, sim.scheme)   # the REFERENCE's executor
    dt = time.time() - t0
    meta["reference_cpu_seconds_per_slice"] = dt
    meta["reference_cpu_threads"] = torch.get_num_threads()
    save_case(os.path.join(HERE, "n53_m20_bigbatch.npz"), tensors, sim.scheme, meta,
              arrays=dict(slice0=res.reshape(-1).numpy().copy()), slicing_indices=fixed)
    print("n53 m20 bigbatch slice 0:", tuple(res.shape), f"{dt:.0f} s", flush=True)


def case_c128_spread():
    """The reference's own complex64-vs-complex128 spread (its executors run in both dtypes on the same
    leaf tensors and schemes): the yardstick for the strict per-amplitude tolerance of SURVEY 8c
    (relative error on |amp| >= 1e-3 rms).  Stores the complex128 results next to the complex64 ones
    in tests/golden/c128_spread.npz plus both error figures per case."""
    from artensor_amd.fixtures import load_case
    out, meta = {}, {}

    def figures(got, want):
        got, want = np.asarray(got).reshape(-1), np.asarray(want).reshape(-1)
        rms = float(np.sqrt(np.mean(np.abs(want) ** 2)))
        loose = float(np.abs(got - want).max() / max(float(np.abs(want).max()), rms))
        sel = np.abs(want) >= 1e-3 * rms
        strict = float((np.abs(got - want)[sel] / np.abs(want)[sel]).max())
        return dict(loose=loose, strict=strict, n=int(sel.sum()))

    for name, sparse in (("n12_dense", False), ("n12_sparse5", True), ("n30_sparse100", True)):
        case = load_case(os.path.join(HERE, name + ".npz"))
        fn = tensor_contraction_sparse if sparse else tensor_contraction
        t0 = time.time()
        r64 = fn(case.fresh_tensors(dtype=torch.complex64), case.scheme).reshape(-1)
        r128 = fn(case.fresh_tensors(dtype=torch.complex128), case.scheme).reshape(-1)
        out[name + "_c128"] = r128.numpy()
        meta[name] = figures(r64.numpy(), r128.numpy())
        print(name, meta[name], f"{time.time() - t0:.0f} s", flush=True)
    np.savez_compressed(os.path.join(HERE, "c128_spread.npz"),
                        meta=np.frombuffer(__import__("json").dumps(meta).encode(), dtype=np.uint8), **out)


def case_random_bench():
    """Benchmark-scale random tensor networks (SURVEY 8d input 2: 3-regular graphs, seed 0, leaves
    complex(randn, randn) / D^1.5, planned by the reference's find_order, trials 4, iters 5):
      * D = 2, 260 vertices, closed, sc_target 30 -> sliced; the reference executor's value of
        slice 0 pins the fixture (about a minute of CPU);
      * D = 4, 100 vertices, closed, sc_target 30 -> no slicing (the reference slice loop is
        only defined for D = 2 bonds); the reference's full value pins it."""
    from artensor_amd.simulation import apply_slice, slice_assignments
    for nv, D, name in [(260, 2, "rand_D2_nv260_sliced"), (100, 4, "rand_D4_nv100")]:
        tensors, tensor_bonds, bond_dims = _rand_tn(nv, D, 0, 0)
        order, slicing_bonds, ctree = find_order(
            deepcopy(tensor_bonds), deepcopy(bond_dims), [], 0, 1, sc_target=30,
            trials=4, iters=5, betas=np.linspace(3.0, 21.0, 61), start_seed=0, slicing_repeat=1)
        scheme, output_bonds = contraction_scheme(deepcopy(ctree))
        assert list(output_bonds) == []
        slicing_indices = {}
        for bond in slicing_bonds:
            slicing_indices[bond] = [(tid, tensor_bonds[tid].index(bond))
                                     for tid in tensor_bonds if bond in tensor_bonds[tid]]
        tc, sc = ctree.tree_complexity()[:2]
        cfg = slice_assignments(len(slicing_bonds), 0)
        sliced = apply_slice(dict(tensors), slicing_indices, cfg) if slicing_bonds else dict(tensors)
        t0 = time.time()
        res = tensor_contraction(sliced, scheme)
        dt = time.time() - t0
        meta = dict(D=D, nv=nv, n_open=0, sc_target=30, log10_tc=float(tc), sc=float(sc), bond_dim=D,
                    output_bonds=[], n_slicing=len(slicing_bonds), reference_cpu_seconds_per_slice=dt,
                    graph="networkx.random_regular_graph(3, nv, seed=0)")
        save_case(os.path.join(HERE, name + ".npz"), tensors, scheme, meta,
                  arrays=dict(slice0=res.reshape(-1).numpy().copy()), slicing_indices=slicing_indices)
        print(name, "steps", len(scheme), "sliced bonds", len(slicing_bonds), "log10 tc/slice", float(tc), "sc", float(sc),
              "slice0", res.reshape(-1), f"{dt:.0f} s", flush=True)


def case_random_bench_nonpow2():
    """Benchmark-scale random 3-regular networks whose bond dimension is NOT a power of two (round 5: the class of
    inputs that used to run on the strided kernel).  Same recipe as case_random_bench (networkx.random_regular_graph(3,
    nv, seed=0), leaves complex(randn, randn) / D^1.5, find_order trials 4, iters 5), sc_target 32 so that the planner
    does not slice (the reference's slice loop enumerates 2**len(bonds) slices: bond dimension 2 only):
      * D = 3, 112 vertices: largest intermediate 3^18 elements (3.1 GB), 10^11.2 complex multiply-adds;
      * D = 6 = 2 x 3, 64 vertices: a composite extent.
    The reference executor's own complex64 value pins each fixture; a complex128 run of the same executor is stored
    beside it (`exact128`: the truth the 1e-5 contract is checked against)."""
    for nv, D, name in [(112, 3, "rand_D3_nv112"), (64, 6, "rand_D6_nv64")]:
        tensors, tensor_bonds, bond_dims = _rand_tn(nv, D, 0, 0)
        order, slicing_bonds, ctree = find_order(
            deepcopy(tensor_bonds), deepcopy(bond_dims), [], 0, 1, sc_target=32,
            trials=4, iters=5, betas=np.linspace(3.0, 21.0, 61), start_seed=0, slicing_repeat=1)
        assert len(slicing_bonds) == 0, (name, slicing_bonds)
        scheme, output_bonds = contraction_scheme(deepcopy(ctree))
        assert list(output_bonds) == []
        tc, sc = ctree.tree_complexity()[:2]
        t0 = time.time()
        res = tensor_contraction(dict(tensors), scheme)
        dt = time.time() - t0
        t0 = time.time()
        res128 = tensor_contraction({i: t.to(torch.complex128) for i, t in tensors.items()}, scheme)
        dt128 = time.time() - t0
        meta = dict(D=D, nv=nv, n_open=0, sc_target=32, log10_tc=float(tc), sc=float(sc), bond_dim=D,
                    output_bonds=[], n_slicing=0, reference_cpu_seconds=dt, reference_cpu_seconds_c128=dt128,
                    graph="networkx.random_regular_graph(3, nv, seed=0)")
        save_case(os.path.join(HERE, name + ".npz"), tensors, scheme, meta,
                  arrays=dict(slice0=res.reshape(-1).numpy().copy(), exact128=res128.reshape(-1).numpy().copy()))
        print(name, "steps", len(scheme), "log10 tc", float(tc), "sc", float(sc), "value", res.reshape(-1), "c128",
              res128.reshape(-1), f"{dt:.0f} s / {dt128:.0f} s", flush=True)


def case_random_bench_nonpow2_open():
    """The OPEN twins of case_random_bench_nonpow2 (round 6): the closed benchmark fixtures pin ONE
    complex scalar each, so a layout slip that cancels in the final dot product would pass.  Same graphs, seeds and
    planner budget, with dangling bonds on the first tensors (tests/golden/make_golden.py::_rand_tn):
      * D = 3, 112 vertices, 6 open bonds: 3^6 = 729 amplitudes;
      * D = 6, 64 vertices, 4 open bonds: 6^4 = 1 296 amplitudes.
    `final` is the reference executor's complex64 result in the order of the scheme's output bonds, `exact128` the
    same executor on complex128 leaves (the truth the 1e-5 contract is checked against)."""
    for nv0, D, n_open, stem in [(112, 3, 6, "rand_D3_open6"), (64, 6, 4, "rand_D6_open4")]:
        # (open bonds raise the space complexity: the largest graph of the family that the planner leaves UNSLICED at
        #  sc_target 30 -- 8 GiB of complex64, 16 of complex128 on this 64 GB host; the reference's slice loop is defined
        #  for bond dimension 2 only)
        for nv in range(nv0, 8, -4):
            tensors, tensor_bonds, bond_dims = _rand_tn(nv, D, 0, n_open)
            order, slicing_bonds, ctree = find_order(
                deepcopy(tensor_bonds), deepcopy(bond_dims), [], 0, 1, sc_target=30,
                trials=4, iters=5, betas=np.linspace(3.0, 21.0, 61), start_seed=0, slicing_repeat=1)
            print(stem, "nv", nv, "sliced bonds", len(slicing_bonds), "sc", float(ctree.tree_complexity()[1]), flush=True)
            if len(slicing_bonds) == 0:
                break
        name = f"{stem}_nv{nv}"
        scheme, output_bonds = contraction_scheme(deepcopy(ctree))
        assert len(output_bonds) == n_open, (name, output_bonds)
        tc, sc = ctree.tree_complexity()[:2]
        t0 = time.time()
        res = tensor_contraction(dict(tensors), scheme)
        dt = time.time() - t0
        t0 = time.time()
        res128 = tensor_contraction({i: t.to(torch.complex128) for i, t in tensors.items()}, scheme)
        dt128 = time.time() - t0
        assert tuple(res.shape) == (D,) * n_open
        meta = dict(D=D, nv=nv, n_open=n_open, sc_target=30, log10_tc=float(tc), sc=float(sc), bond_dim=D,
                    output_bonds=[int(b) for b in output_bonds], n_slicing=0, reference_cpu_seconds=dt,
                    reference_cpu_seconds_c128=dt128, graph="networkx.random_regular_graph(3, nv, seed=0)")
        save_case(os.path.join(HERE, name + ".npz"), tensors, scheme, meta,
                  arrays=dict(final=res.numpy().copy(), exact128=res128.numpy().copy()))
        err = (res.to(torch.complex128) - res128).abs().max() / res128.abs().max()
        print(name, "steps", len(scheme), "log10 tc", float(tc), "sc", float(sc), "amplitudes", res.numel(),
              "c64 vs c128", err.item(), f"{dt:.0f} s / {dt128:.0f} s", flush=True)


def case_gates():
    """Gate lists (array + bond labels per gate, in circuit order) of the n12 and n30 circuits as
    the reference's TensorNetworkCircuit builds them (circuit.py:100-130): the input of
    `state_vec()` (circuit.py:155-175).  n12's state vector is already in n12_dense.npz."""
    for name, qsim in (("n12", N12_QSIM), ("n30", n30_qsim())):
        c = TensorNetworkCircuit(qsim)
        gates = c.circuits_tn[c.n:]
        tensors = {k: g.array.to(torch.complex64) for k, g in enumerate(gates)}
        meta = dict(n_qubits=int(c.n), inds=[list(map(str, g.inds)) for g in gates])
        save_case(os.path.join(HERE, f"{name}_gates.npz"), tensors, [], meta)
        print(name, "gates", len(gates))


CASES = {
    "gates": case_gates,
    "random_bench": case_random_bench,
    "random_bench_nonpow2": case_random_bench_nonpow2,
    "random_bench_nonpow2_open": case_random_bench_nonpow2_open,
    "n53_plan": case_n53_plan,
    "n53_slice0": case_n53_slice0,
    "n53m20_plan": lambda: case_n53_plan("m20"),
    "n53m20_slice0": lambda: case_n53_slice0("m20"),
    "c128_spread": case_c128_spread,
    "n53m20_batch_plan": case_n53m20_batch_plan,
    "n53m20_batch_slice0": case_n53m20_batch_slice0,
    "n53m20_bigbatch": case_n53m20_bigbatch,
    "trees": case_trees,
    "n12_dense": case_n12_dense,
    "n12_sparse5": case_n12_sparse5,
    "n12_sparse_sliced": case_n12_sparse_sliced,
    "n12_scinot": case_n12_scinot,
    "random": case_random_tns,
    "n30_plan": case_n30_plan,
    "n30_run": case_n30_run,
    "n30_sparse100": lambda: case_n30_sparse(100),
    "n30_sparse10000": lambda: case_n30_sparse(10000),
    "n30_sliced3": lambda: case_n30_sliced(3),
    "n30_dense_parts": case_n30_dense_parts,
}

if __name__ == "__main__":
    todo = sys.argv[1:] or ["n12_dense", "n12_sparse5", "n12_sparse_sliced", "n12_scinot", "random"]
    for name in todo:
        t0 = time.time()
        CASES[name]()
        print(f"[{name}] {time.time() - t0:.1f} s", flush=True)
