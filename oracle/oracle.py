"""CPU oracle: a numpy restatement of the reference's numerical executors.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import it; artensor_amd never does.

It restates, function by function, what the reference computes on the hot path:

  tensor_contraction          /root/reference/artensor/contraction.py:62-76
  tensor_contraction_sparse   /root/reference/artensor/contraction.py:132-205
  sliced_contraction          /root/reference/artensor/simulation.py:101-116 (twin :198-213)

The reference's arithmetic is `torch.einsum` (requirements: pytorch, examples pin 1.12.1);
here every pairwise step is lowered by hand to transpose + reshape + complex matmul in
numpy, so the oracle shares no code with either torch.einsum or the HIP kernels.

Pinning (parity is NOT "unpinned"): tests/test_oracle.py checks this file against the
golden fixtures under tests/golden/, which tests/golden/make_golden.py produced by running
the reference itself in the build container -- n12/n30 dense, n12/n30 sparse, sliced and
random-network cases, including the reference's own known-answer table
(/root/reference/tests/test_circuits.py:25-31) and Google's n30 amplitudes
(/root/reference/examples/amplitudes_n30_m14_s0_e0_pEFGH_10000.txt).
"""
import numpy as np


def parse_eq(eq):
    """'ab,bc->ac' -> (list a, list b, list out).  Labels are single characters
    (reference alphabet A-Y, a-y: contraction.py:9-10)."""
    if not isinstance(eq, str):   # a triple of label tuples (any hashable labels)
        return list(eq[0]), list(eq[1]), list(eq[2])
    lhs, out = eq.split("->")
    a, b = lhs.split(",")
    return list(a), list(b), list(out)


def einsum_pair(eq, a, b):
    """One pairwise step, i.e. torch.einsum(eq, a, b) of contraction.py:70.

    Classify labels as batch (in a, b and out), contracted (in a and b, not out), free-a,
    free-b, summed-alone (in one operand only and not in out); bring a to
    [batch, free_a, contracted], b to [batch, contracted, free_b]; matmul; permute the
    [batch, free_a, free_b] result into the requested output order."""
    la, lb, lo = parse_eq(eq)
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.ndim == len(la) and b.ndim == len(lb), (eq, a.shape, b.shape)
    # labels living in a single operand and not in the output are summed out first
    for lab in [x for x in la if x not in lb and x not in lo]:
        ax = la.index(lab)
        a = a.sum(axis=ax)
        la.pop(ax)
    for lab in [x for x in lb if x not in la and x not in lo]:
        ax = lb.index(lab)
        b = b.sum(axis=ax)
        lb.pop(ax)
    batch = [x for x in la if x in lb and x in lo]
    contr = [x for x in la if x in lb and x not in lo]
    free_a = [x for x in la if x not in lb]
    free_b = [x for x in lb if x not in la]
    dim = {}
    for lab, n in zip(la, a.shape):
        dim[lab] = n
    for lab, n in zip(lb, b.shape):
        assert dim.setdefault(lab, n) == n, (eq, lab)
    prod = lambda labs: int(np.prod([dim[x] for x in labs], dtype=np.int64)) if labs else 1
    a2 = a.transpose([la.index(x) for x in batch + free_a + contr]).reshape(
        prod(batch), prod(free_a), prod(contr))
    b2 = b.transpose([lb.index(x) for x in batch + contr + free_b]).reshape(
        prod(batch), prod(contr), prod(free_b))
    c2 = np.matmul(a2, b2)
    cur = batch + free_a + free_b
    c = c2.reshape([dim[x] for x in cur])
    assert sorted(cur) == sorted(lo), (eq, cur, lo)
    return c.transpose([cur.index(x) for x in lo]).copy(order="C")


def tensor_contraction(tensors, scheme):
    """contraction.py:62-76: tensors[i] <- einsum(eq, tensors[i], tensors[j]) per step;
    returns the last tensors[i].  `tensors` (dict or list) is mutated like the reference."""
    i = None
    for step in scheme:
        (i, j), eq = step[0], step[1]
        tensors[i] = einsum_pair(eq, tensors[i], tensors[j])
    return tensors[i]


def _idx(x):
    return np.asarray(x, dtype=np.int64)


def tensor_contraction_sparse(tensors, scheme, scientific_notation=False):
    """contraction.py:132-205.  The four branches, keyed exactly like the reference:

    (A) len(batch_i) > 1  : chunk loop -- gather rows of both operands per chunk, batched
        einsum, optional reshape(step[3]), concatenate along dim 0       (:140-175)
    (B) 5-tuple, one index list per operand: gather both, batched einsum  (:176-179)
    (C) 5-tuple otherwise: einsum, reshape(step[3]), optional row select  (:180-188)
    (D) 3-tuple: plain einsum                                             (:189-191)
    then optionally divide by abs().max() and accumulate log10            (:197-200)."""
    factor = 0.0
    i = None
    for step in scheme:
        i, j = step[0]
        eq = step[1]
        batch_i, batch_j = step[2]
        if len(batch_i) > 1:
            parts = []
            for k in range(len(batch_i)):
                r = einsum_pair(eq, tensors[i][_idx(batch_i[k])], tensors[j][_idx(batch_j[k])])
                if step[3]:
                    r = r.reshape(step[3])
                parts.append(r)
            tensors[j] = []
            tensors[i] = np.concatenate(parts, axis=0)
        elif len(step) > 3 and len(batch_i) == len(batch_j) == 1:
            tensors[i] = tensors[i][_idx(batch_i[0])]
            tensors[j] = tensors[j][_idx(batch_j[0])]
            tensors[i] = einsum_pair(eq, tensors[i], tensors[j])
        elif len(step) > 3:
            tensors[i] = einsum_pair(eq, tensors[i], tensors[j]).reshape(step[3])
            if len(batch_i) == 1:
                tensors[i] = tensors[i][_idx(batch_i[0])]
            tensors[j] = []
        else:
            tensors[i] = einsum_pair(eq, tensors[i], tensors[j])
            tensors[j] = []
        if scientific_notation:
            norm = np.abs(tensors[i]).max()
            tensors[i] = tensors[i] / norm
            factor += np.log10(norm)
    if scientific_notation:
        return factor, tensors[i]
    return tensors[i]


def sliced_contraction(tensors, scheme, slicing_indices, out_shape, sparse=False,
                       permute_dims=None, dtype=np.complex64, slices=None):
    """The slice loop of simulation.py:101-116: for every assignment of the sliced bonds
    (MSB = first bond, :108) select that index on every tensor carrying the bond, run the
    executor, accumulate; finally permute.  `slicing_indices` is the ordered mapping
    bond -> [(tensor_id, dim_index), ...] computed on the UNSLICED tensors (:62-65).

    The reference applies the selects bond after bond with those stale dim indices, which
    is only well defined when no tensor carries two sliced bonds in ascending dim order
    (SURVEY.md 8a row S); like the reference, this restatement does not repair that --
    fixtures are generated only for well-defined slicings.

    `slices` optionally restricts the loop to a subset of slice numbers (what one rank of
    a multi-GPU run computes)."""
    execute = tensor_contraction_sparse if sparse else tensor_contraction
    bonds = list(slicing_indices.keys()) if slicing_indices else []
    collect = np.zeros(out_shape, dtype=dtype)
    todo = range(2 ** len(bonds)) if slices is None else slices
    for s in todo:
        cfg = [int(c) for c in np.binary_repr(s, len(bonds))] if bonds else []
        sliced = dict(tensors) if isinstance(tensors, dict) else list(tensors)
        for x, bond in enumerate(bonds):
            for tid, ind in slicing_indices[bond]:
                sliced[tid] = np.take(sliced[tid], cfg[x], axis=ind).copy(order="C")
        collect += execute(sliced, scheme)
    if permute_dims is not None and len(permute_dims):
        collect = collect.transpose(permute_dims)
    return collect


def scheme_flops(tensors, scheme):
    """Real FLOP of a dense scheme: 8 * prod(extent of every label of the step), summed
    (equals 8 * 10**tc of contraction_tree.py:415-452)."""
    shapes = {k: tuple(np.asarray(v).shape) for k, v in
              (tensors.items() if isinstance(tensors, dict) else enumerate(tensors))}
    total = 0.0
    per_step = []
    for step in scheme:
        (i, j), eq = step[0], step[1]
        la, lb, lo = parse_eq(eq)
        dim = dict(zip(la, shapes[i]))
        dim.update(zip(lb, shapes[j]))
        f = 8.0 * float(np.prod([float(dim[x]) for x in dim]))
        per_step.append(f)
        total += f
        shapes[i] = tuple(dim[x] for x in lo)
    return total, per_step


def tensor_contraction_torch_cpu(tensors, scheme, budget_s=None, threads=None):
    """The reference's dense executor as it runs on a CPU: `tensors[i] = torch.einsum(eq, tensors[i],
    tensors[j])` for every step (/root/reference/artensor/contraction.py:62-76), on torch-CPU tensors.
    This is the CPU baseline SURVEY 8d specifies (bench.py's cpu_baseline leg); nothing else calls it.

    budget_s: stop after the first step that ends past this many seconds (a bounded sample of a big
    scheme).  Returns dict(result (None if stopped early), steps_done, flops_done, flops_total,
    seconds, threads)."""
    import time
    import torch
    if threads:
        torch.set_num_threads(int(threads))
    ts = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))) for k, v in
          (tensors.items() if isinstance(tensors, dict) else enumerate(tensors))}
    shapes = {k: tuple(t.shape) for k, t in ts.items()}
    per_step = []
    for (i, j), eq in scheme:
        a, b, out = parse_eq(eq)
        ext = dict(zip(a, shapes[i]))
        ext.update(zip(b, shapes[j]))
        f = 8.0
        for x in ext.values():
            f *= x
        per_step.append(f)
        shapes[i] = tuple(ext[x] for x in out)
    done_f, done_n = 0.0, 0
    t0 = time.perf_counter()
    for n, ((i, j), eq) in enumerate(scheme):
        ts[i] = torch.einsum(eq, ts[i], ts[j])
        done_f += per_step[n]
        done_n += 1
        if budget_s is not None and time.perf_counter() - t0 > budget_s and n + 1 < len(scheme):
            break
    dt = time.perf_counter() - t0
    whole = done_n == len(scheme)
    return dict(result=ts[scheme[-1][0][0]] if whole else None, steps_done=done_n, flops_done=done_f,
                flops_total=float(sum(per_step)), seconds=dt, threads=torch.get_num_threads())


def tensor_contraction_sparse_torch_cpu(tensors, scheme, budget_s=None, threads=None):
    """The reference's sparse-state executor as it runs on a CPU (/root/reference/artensor/contraction.py:132-205,
    restated branch by branch on torch-CPU tensors: chunk loop with row gathers and torch.cat :140-175, gathered
    batched einsum :176-179, einsum + reshape + optional row select :180-188, plain einsum :189-191).  CPU baseline
    of bench.py's sparse and sliced workloads; nothing else calls it.  budget_s / return value as
    tensor_contraction_torch_cpu (FLOP of a step = 8 * product of the extents of every label of its einsum, summed
    over the chunks of a chunked step)."""
    import time
    import torch
    if threads:
        torch.set_num_threads(int(threads))
    ts = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))) for k, v in
          (tensors.items() if isinstance(tensors, dict) else enumerate(tensors))}

    def flops_of(eq, a, b):
        la, lb, _ = parse_eq(eq)
        ext = dict(zip(la, a.shape))
        ext.update(zip(lb, b.shape))
        f = 8.0
        for x in ext.values():
            f *= x
        return f

    done_f, done_n = 0.0, 0
    t0 = time.perf_counter()
    for n, step in enumerate(scheme):
        i, j = step[0]
        eq = step[1]
        batch_i, batch_j = step[2]
        if len(batch_i) > 1:
            parts = []
            for k in range(len(batch_i)):
                a, b = ts[i][batch_i[k]], ts[j][batch_j[k]]
                done_f += flops_of(eq, a, b)
                r = torch.einsum(eq, a, b)
                if step[3]:
                    r = r.reshape(step[3])
                parts.append(r)
            ts[j] = []
            ts[i] = torch.cat(parts, dim=0)
        elif len(step) > 3 and len(batch_i) == len(batch_j) == 1:
            ts[i] = ts[i][batch_i[0]]
            ts[j] = ts[j][batch_j[0]]
            done_f += flops_of(eq, ts[i], ts[j])
            ts[i] = torch.einsum(eq, ts[i], ts[j])
        elif len(step) > 3:
            done_f += flops_of(eq, ts[i], ts[j])
            ts[i] = torch.einsum(eq, ts[i], ts[j]).reshape(step[3])
            if len(batch_i) == 1:
                ts[i] = ts[i][batch_i[0]]
            ts[j] = []
        else:
            done_f += flops_of(eq, ts[i], ts[j])
            ts[i] = torch.einsum(eq, ts[i], ts[j])
            ts[j] = []
        done_n += 1
        if budget_s is not None and time.perf_counter() - t0 > budget_s and n + 1 < len(scheme):
            break
    dt = time.perf_counter() - t0
    whole = done_n == len(scheme)
    return dict(result=ts[scheme[-1][0][0]] if whole else None, steps_done=done_n, flops_done=done_f,
                flops_total=None, seconds=dt, threads=torch.get_num_threads())


# The two torch loops above run on whatever device their tensors live on.  With complex128 CUDA tensors they are the
# reference's executors as they run on a GPU (torch.einsum -> permute + bmm on the vendor BLAS): an independent complex128
# truth for the cases whose 2^30-element intermediates (16 GiB each in complex128) do not fit the build container
# (tests/test_gpu_parity.py::test_c128_truth_against_torch_einsum_on_the_gpu).  Test infrastructure only.
tensor_contraction_torch = tensor_contraction_torch_cpu
tensor_contraction_sparse_torch = tensor_contraction_sparse_torch_cpu
