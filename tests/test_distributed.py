"""The N > 1 path on CPU: slice sharding + the single reduce of artensor_amd.sliced_contraction
with world_size 2 over gloo.  The HIP executors cannot run here, so the test builds its runner with
SliceRunner._with_seams (CPU oracle as executor) and enters below the public signature at
_shard_and_reduce; what is exercised is the product's slice assignment, per-tensor slice
application, accumulation order and the collective.  The same path with the real HIP executors
and two processes on one GPU is tests/test_gpu_parity.py::test_two_process_sliced_contraction."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import artensor_amd as A
from artensor_amd.fixtures import load_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_execute(sparse):
    from oracle import oracle

    def run(tensors, scheme):
        np_t = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in tensors.items()}
        fn = oracle.tensor_contraction_sparse if sparse else oracle.tensor_contraction
        return torch.from_numpy(np.ascontiguousarray(fn(np_t, scheme)))
    return run


def _cpu_add(acc, x):
    acc += x
    return acc


def _worker(rank, world, port, name, reduce, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case = load_case(os.path.join(GOLDEN, name + ".npz"))
        want = case.arrays["final"]
        sparse = case.meta.get("pattern") == "sparse"
        from artensor_amd import simulation as S
        runner = S.SliceRunner._with_seams(case.tensors, case.scheme, case.slicing_indices, want.shape, sparse,
                                           torch.complex64, "cpu", _oracle_execute(sparse), _cpu_add)
        out = S._shard_and_reduce(runner, reduce=reduce)
        np.save(os.path.join(out_dir, f"{name}_{reduce}_{rank}.npy"), out.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,reduce", [("n12_sparse_sliced", "all"), ("rand_D2_closed_sliced", "root"),
                                         ("rand_D2_open_sliced", "all")])
def test_two_ranks_shard_slices_and_reduce_once(tmp_path, name, reduce):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, name, reduce, str(tmp_path)), nprocs=2, join=True)
    case = load_case(os.path.join(GOLDEN, name + ".npz"))
    want = case.arrays["final"]
    r0 = np.load(tmp_path / f"{name}_{reduce}_0.npy")
    r1 = np.load(tmp_path / f"{name}_{reduce}_1.npy")
    scale = np.abs(want).max()
    assert np.abs(r0 - want).max() <= 1e-5 * scale          # rank 0 always holds the full sum
    if reduce == "all":
        assert np.abs(r1 - want).max() <= 1e-5 * scale      # all_reduce: so does rank 1
    else:
        assert np.abs(r1 - want).max() > 1e-3 * scale        # reduce-to-root: rank 1 keeps its partial sum


def test_slice_bookkeeping():
    assert A.slice_assignments(3, 5) == [1, 0, 1]            # MSB = first bond (simulation.py:108)
    assert A.slice_assignments(0, 0) == []
    assert list(A.rank_slices(8, 1, 4)) == [1, 5]
    assert sorted(sum((list(A.rank_slices(8, r, 3)) for r in range(3)), [])) == list(range(8))
    # one tensor carrying two sliced bonds in ASCENDING dim order: the reference's
    # bond-by-bond select() goes stale here (simulation.py:110-113); this package indexes
    # the unsliced tensor once
    t = torch.arange(24).reshape(2, 3, 4)
    out = A.apply_slice({0: t, 1: t}, {"x": [(0, 0)], "y": [(0, 2)]}, [1, 3])
    assert torch.equal(out[0], t[1, :, 3]) and out[1] is t
    # descending order is where the reference is well defined: same answer as its selects
    out = A.apply_slice({0: t}, {"y": [(0, 2)], "x": [(0, 0)]}, [3, 1])
    assert torch.equal(out[0], t.select(2, 3).select(0, 1))


def _worker_mismatch(rank, world, port, out_dir):
    """Rank 1 holds a different slicing of the same network (what PYTHONHASHSEED does to the reference's planner):
    the fingerprint check of _shard_and_reduce must refuse to sum the slices."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case = load_case(os.path.join(GOLDEN, "n12_sparse_sliced.npz"))
        slicing = dict(case.slicing_indices)
        if rank == 1:   # the same bonds in the other order: slice number s means another assignment
            slicing = dict(reversed(list(slicing.items())))
        from artensor_amd import simulation as S
        runner = S.SliceRunner._with_seams(case.tensors, case.scheme, slicing, case.arrays["final"].shape, True,
                                           torch.complex64, "cpu", _oracle_execute(True), _cpu_add)
        try:
            S._shard_and_reduce(runner, reduce="all")
            msg = "no error"
        except RuntimeError as e:
            msg = str(e)
        with open(os.path.join(out_dir, f"mismatch_{rank}.txt"), "w") as f:
            f.write(msg)
    finally:
        dist.destroy_process_group()


def test_ranks_with_different_plans_are_refused(tmp_path):
    port = _free_port()
    mp.spawn(_worker_mismatch, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert "different plans" in open(tmp_path / f"mismatch_{r}.txt").read()


class _FakePlanner:
    """A planner module whose products depend on the rank (the reference's do, through PYTHONHASHSEED): only what
    rank 0 plans may be executed."""
    def __init__(self, case, rank):
        self.case, self.rank = case, rank

    def NumericalTensorNetwork(self, tensors, tensor_bonds, bond_dims, final_qubits):
        case, rank = self.case, self.rank

        class Net:
            pass
        net = Net()
        net.tensors = dict(tensors)
        net.bond_dims = bond_dims
        net._simplify = lambda pattern: (tensor_bonds, final_qubits)
        return net

    def find_order(self, *a, **k):
        return None, list(self.case.slicing_indices.keys()), ("tree of rank", self.rank)


def _worker_plan_once(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from artensor_amd import simulation as S
        from artensor_amd import contraction as C
        case = load_case(os.path.join(GOLDEN, "rand_D2_closed_sliced.npz"))
        planned = []
        # the scheme compiler stands in for the tree walk: it records which rank's tree it was given
        orig = C.contraction_scheme
        C.contraction_scheme = lambda ctree, labels="einsum": (planned.append(ctree) or (case.scheme, []))
        seen = {}
        orig_sc = S.sliced_contraction
        S.sliced_contraction = lambda leaves, scheme, slicing, shape, **kw: seen.update(n=len(scheme), bonds=list(slicing)) or torch.zeros(())
        try:
            bonds = {k: [f"b{k}_{d}" for d in range(t.dim())] for k, t in case.tensors.items()}
            for bond, lst in case.slicing_indices.items():   # name the sliced bonds as the fixture does
                for tid, dim in lst:
                    bonds[tid][dim] = bond
            S.tensor_network_contraction(case.tensors, bonds, {}, [], planner=_FakePlanner(case, rank), device="cpu")
        finally:
            C.contraction_scheme, S.sliced_contraction = orig, orig_sc
        with open(os.path.join(out_dir, f"plan_{rank}.txt"), "w") as f:
            f.write(f"{len(planned)} {seen['n']} {seen['bonds']}")
    finally:
        dist.destroy_process_group()


def test_one_rank_plans_and_the_plan_is_broadcast(tmp_path):
    """tensor_network_contraction under torch.distributed: rank 0 runs the planner and the scheme compiler, every
    other rank receives the plan (leaves, scheme, slicing, output order) and plans nothing itself."""
    port = _free_port()
    mp.spawn(_worker_plan_once, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (open(tmp_path / f"plan_{r}.txt").read().split(" ", 2) for r in range(2))
    assert r0[0] == "1" and r1[0] == "0"          # only rank 0 compiled a scheme
    assert r0[1:] == r1[1:]                        # both execute the same scheme on the same slicing


def _worker_plan_fails(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from artensor_amd import simulation as S
        from artensor_amd import contraction as C
        case = load_case(os.path.join(GOLDEN, "rand_D2_closed_sliced.npz"))
        orig = C.contraction_scheme

        def boom(ctree, labels="einsum"):
            raise ValueError("planner exploded")
        C.contraction_scheme = boom
        try:
            bonds = {k: [f"b{k}_{d}" for d in range(t.dim())] for k, t in case.tensors.items()}
            try:
                S.tensor_network_contraction(case.tensors, bonds, {}, [], planner=_FakePlanner(case, rank), device="cpu")
                msg = "no error"
            except RuntimeError as e:
                msg = str(e)
        finally:
            C.contraction_scheme = orig
        with open(os.path.join(out_dir, f"fail_{rank}.txt"), "w") as f:
            f.write(msg)
    finally:
        dist.destroy_process_group()


def test_a_planner_failure_on_rank_0_raises_on_every_rank(tmp_path):
    """only rank 0 plans; if it raises, the other ranks must not wait in the broadcast for ever -- the error
    travels in the plan's place and every rank raises RuntimeError."""
    port = _free_port()
    mp.spawn(_worker_plan_fails, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        msg = open(tmp_path / f"fail_{r}.txt").read()
        assert "planning failed on rank 0" in msg and "planner exploded" in msg, (r, msg)


def test_checkpoint_and_resume_of_the_slice_loop(tmp_path, monkeypatch):
    """SURVEY section 5 (optional hook): the slice loop writes its partial sum every few slices and a later call resumes after
    the last saved slice -- same result as the uninterrupted loop; a checkpoint of another plan or another shard is refused.
    (Single process, CPU oracle through the test seams: what is exercised is the product's checkpoint logic.)"""
    from artensor_amd import simulation as S
    case = load_case(os.path.join(GOLDEN, "rand_D2_closed_sliced.npz"))     # 6 sliced bonds: 64 slices
    want = case.arrays["final"]

    def runner():
        return S.SliceRunner._with_seams(case.tensors, case.scheme, case.slicing_indices, want.shape, False,
                                         torch.complex64, "cpu", _oracle_execute(False), _cpu_add)
    full = S._shard_and_reduce(runner(), reduce=None).numpy().copy()
    assert np.abs(full - want).max() <= 1e-5 * np.abs(want).max()
    prefix = str(tmp_path / "ck")
    # an interrupted run: the 4th save never happens
    saves = []
    orig = S.save_checkpoint

    def dying(path, fp, sl, done, collect, digest=None):
        if len(saves) == 3:
            raise KeyboardInterrupt
        saves.append(done)
        orig(path, fp, sl, done, collect, digest)
    monkeypatch.setattr(S, "save_checkpoint", dying)
    with pytest.raises(KeyboardInterrupt):
        S._shard_and_reduce(runner(), reduce=None, checkpoint=prefix, checkpoint_every=10)
    assert saves == [10, 20, 30] and os.path.exists(prefix + ".rank0of1.pt") and not os.path.exists(prefix + ".rank0of1.pt.tmp")
    monkeypatch.setattr(S, "save_checkpoint", orig)
    # resumed: only slices 30.. are contracted again
    r = runner()
    ran = []
    orig_run = r.run
    r.run = lambda sl: (ran.append(list(sl)), orig_run(sl))[1]
    resumed = S._shard_and_reduce(r, reduce=None, checkpoint=prefix, checkpoint_every=10).numpy()
    assert [len(x) for x in ran] == [10, 10, 10, 4] and sum(len(x) for x in ran) == 64 - 30
    assert np.abs(resumed - full).max() <= 1e-6 * np.abs(full).max()
    # a finished checkpoint resumes to "nothing left": no slice is contracted twice
    r = runner()
    r.run = lambda sl: (_ for _ in ()).throw(AssertionError("nothing should run"))
    again = S._shard_and_reduce(r, reduce=None, checkpoint=prefix, checkpoint_every=10).numpy()
    assert np.abs(again - full).max() <= 1e-6 * np.abs(full).max()
    # another shard of the same plan, and another plan: refused
    with pytest.raises(RuntimeError, match="another shard"):
        S._shard_and_reduce(runner(), reduce=None, slices=list(range(5, 40)), checkpoint=prefix, checkpoint_every=10)
    other = load_case(os.path.join(GOLDEN, "rand_D2_open_sliced.npz"))
    r2 = S.SliceRunner._with_seams(other.tensors, other.scheme, other.slicing_indices, other.arrays["final"].shape, False,
                                   torch.complex64, "cpu", _oracle_execute(False), _cpu_add)
    with pytest.raises(RuntimeError, match="another plan"):
        S._shard_and_reduce(r2, reduce=None, checkpoint=prefix, checkpoint_every=10)
    # the SAME plan with other leaf values (same circuit structure, other gate parameters), another dtype: refused --
    # a finished file would otherwise be returned as the new run's result without contracting anything
    scaled = {k: t * 2 for k, t in case.tensors.items()}
    r3 = S.SliceRunner._with_seams(scaled, case.scheme, case.slicing_indices, want.shape, False, torch.complex64, "cpu",
                                   _oracle_execute(False), _cpu_add)
    with pytest.raises(RuntimeError, match="other leaf tensors"):
        S._shard_and_reduce(r3, reduce=None, checkpoint=prefix, checkpoint_every=10)
    r4 = S.SliceRunner._with_seams(case.tensors, case.scheme, case.slicing_indices, want.shape, False, torch.complex128, "cpu",
                                   _oracle_execute(False), _cpu_add)
    with pytest.raises(RuntimeError, match="other leaf tensors|partial sum"):
        S._shard_and_reduce(r4, reduce=None, checkpoint=prefix, checkpoint_every=10)
    # same length, same tail, other slices: the whole list is hashed
    perm = list(range(64))
    perm[0], perm[1] = perm[1], perm[0]
    with pytest.raises(RuntimeError, match="slice list"):
        S._shard_and_reduce(runner(), reduce=None, slices=perm, checkpoint=prefix, checkpoint_every=10)


# ---------------------------------------------------------------------------------------------
# world sizes 3 and 4: a shard count that is not a power of two, a sub-group that does not
# contain global rank 0, and one rank raising in the middle of its slice loop
# ---------------------------------------------------------------------------------------------
def _worker_wide(rank, world, port, name, mode, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case = load_case(os.path.join(GOLDEN, name + ".npz"))
        want = case.arrays["final"]
        sparse = case.meta.get("pattern") == "sparse"
        from artensor_amd import simulation as S
        execute = _oracle_execute(sparse)
        if mode == "raise" and rank == world - 1:
            calls = [0]
            inner = execute

            def execute(tensors, scheme):   # the last rank's SECOND slice fails
                calls[0] += 1
                if calls[0] == 2:
                    raise ValueError("slice went wrong on purpose")
                return inner(tensors, scheme)
        runner = S.SliceRunner._with_seams(case.tensors, case.scheme, case.slicing_indices, want.shape, sparse,
                                           torch.complex64, "cpu", execute, _cpu_add)
        group = None
        if mode == "subgroup":      # every rank creates the group (a collective); only its members use it
            group = dist.new_group(ranks=list(range(1, world)))
            if rank == 0:
                return
        try:
            out = S._shard_and_reduce(runner, group=group, reduce="root" if mode == "subgroup" else "all")
            np.save(os.path.join(out_dir, f"{mode}_{rank}.npy"), out.numpy())
        except Exception as e:
            with open(os.path.join(out_dir, f"{mode}_{rank}.txt"), "w") as f:
                f.write(f"{type(e).__name__}: {e}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [3, 4])
def test_three_and_four_ranks_shard_the_slices(tmp_path, world):
    """64 slices over 3 ranks (22 + 21 + 21: shards whose count is not a power of two) and over 4."""
    name = "rand_D2_closed_sliced"
    port = _free_port()
    mp.spawn(_worker_wide, args=(world, port, name, "all", str(tmp_path)), nprocs=world, join=True)
    want = load_case(os.path.join(GOLDEN, name + ".npz")).arrays["final"]
    for r in range(world):
        got = np.load(tmp_path / f"all_{r}.npy")
        assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()


def test_a_sub_group_without_global_rank_0_reduces_to_its_own_root(tmp_path):
    """reduce="root" inside a group made of global ranks 1 and 2: the sum lands on global rank 1 (group rank 0).  torch's
    dist.reduce takes a GLOBAL destination rank; with dst=0 this raised (rank 0 is not in the group)."""
    name = "rand_D2_closed_sliced"
    port = _free_port()
    mp.spawn(_worker_wide, args=(3, port, name, "subgroup", str(tmp_path)), nprocs=3, join=True)
    want = load_case(os.path.join(GOLDEN, name + ".npz")).arrays["final"]
    assert not (tmp_path / "subgroup_1.txt").exists(), open(tmp_path / "subgroup_1.txt").read()
    root = np.load(tmp_path / "subgroup_1.npy")
    other = np.load(tmp_path / "subgroup_2.npy")
    assert np.abs(root - want).max() <= 1e-5 * np.abs(want).max()
    assert np.abs(other - want).max() > 1e-3 * np.abs(want).max()     # keeps its partial sum


def test_a_slice_that_raises_on_one_rank_raises_on_every_rank(tmp_path):
    """No checkpoint: rank 2's second slice raises.  Ranks 0 and 1 must not be left waiting in the all_reduce
    (mp.spawn would hang until pytest's timeout); every rank raises, the failing one its own exception."""
    port = _free_port()
    mp.spawn(_worker_wide, args=(3, port, "rand_D2_closed_sliced", "raise", str(tmp_path)), nprocs=3, join=True)
    msgs = [open(tmp_path / f"raise_{r}.txt").read() for r in range(3)]
    assert msgs[2].startswith("ValueError: slice went wrong on purpose")
    for r in (0, 1):
        assert "the slice loop failed on rank 2" in msgs[r] and "on purpose" in msgs[r]
