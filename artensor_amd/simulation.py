"""Driver side of the hot path: the slice loop of the reference's
`TensorNetworkSimulation.contraction` (artensor/simulation.py:90-117, twin :198-213),
sharded over the GPUs of one node.

Every slice (one assignment of the sliced bonds) runs the same scheme on leaf tensors
with those bonds fixed; slices are independent, so rank r takes slices r, r+W, r+2W, ...
and accumulates a local `collect_tensor` with the HIP axpy kernel.  The only exchange is
ONE sum of `collect_tensor` across ranks at the end (`torch.distributed` reduce; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests of the sharding logic).
"""
import ctypes

import numpy as np
import torch

from . import _native as N
from . import contraction as _C
from .contraction import _labels, _sparse_step, contract, tensor_contraction, tensor_contraction_sparse

__all__ = ["slice_assignments", "rank_slices", "apply_slice", "accumulate", "sliced_contraction",
           "SliceRunner", "TensorNetworkSimulation", "tensor_network_contraction", "quantum_circuit_simulation",
           "partition_output", "partitioned_contraction"]


def slice_assignments(n_bonds, s):
    """Bits of slice number s, MSB = first sliced bond (reference simulation.py:108)."""
    return [int(c) for c in np.binary_repr(s, n_bonds)] if n_bonds else []


def rank_slices(n_slices, rank, world_size, gray=False):
    """Slices of `rank`: round-robin (s = rank + world_size * t), so every rank gets the same
    count +-1.  gray=True visits the same set with t in reflected-Gray-code order: consecutive
    slices then differ in ONE sliced bond (when the counts are powers of two), which is what
    SliceRunner's reuse of slice-independent intermediate tensors feeds on."""
    count = len(range(rank, n_slices, world_size))
    if not gray:
        return range(rank, n_slices, world_size)
    return _GraySlices(rank, world_size, count)


class _GraySlices:
    """rank + world_size * g for g = t ^ (t >> 1), t = 0, 1, ... (values g >= count skipped): the Gray-ordered shard of
    rank_slices as a LAZY sequence -- a plan with 41 sliced bonds has 2^41 slices, and a list of a rank's share does not fit any
    host (a diagnostic that built one took its machine down in round 5).  Iteration, len(), integer indices and slices
    (a slice is a list: prefixes and the checkpoint loop's pieces are what callers take) cost what they touch: an integer
    index is O(log^2 count) for any count (a rank/select over the reflected Gray sequence, `_below`), never a walk from 0."""
    __slots__ = ("rank", "world", "count", "span")

    def __init__(self, rank, world_size, count):
        self.rank, self.world, self.count = int(rank), int(world_size), int(count)
        span = 1
        while span < self.count:
            span *= 2
        self.span = span

    def __len__(self):
        return self.count

    def _below(self, n, T, c, rev=False):
        """How many of the first T entries of the n-bit reflected Gray sequence (reversed: rev) are < c.  The sequence is
        0.G(n-1) followed by 1.reverse(G(n-1)), so one of the two halves is always whole or absent: O(n)."""
        total = 0
        while True:
            if T <= 0 or c <= 0:
                return total
            size = 1 << n
            if c >= size:
                return total + min(T, size)
            if T >= size:
                return total + c
            half = size >> 1
            # first half of the (possibly reversed) sequence: non-reversed -> values G(n-1) below `half`;
            # reversed -> values half + G(n-1) (the reverse of the reversed upper half)
            if not rev:
                if T <= half:
                    n, c = n - 1, min(c, half)
                    continue
                total += min(c, half)                       # the whole lower half
                if c <= half:
                    return total
                n, T, c, rev = n - 1, T - half, c - half, True
            else:
                if T <= half:
                    if c <= half:
                        return total
                    n, c, rev = n - 1, c - half, False
                    continue
                total += max(0, c - half)                   # the whole upper half (values half .. size-1)
                n, T, c = n - 1, T - half, min(c, half)

    def _t_of(self, pos):
        """the counter value t whose Gray code is the shard's entry number `pos` (O(log^2): a binary search over _below)"""
        if self.count == self.span:
            return pos
        n = self.span.bit_length() - 1
        lo, hi = 0, self.span            # smallest T with _below(T) == pos + 1; t = T - 1
        while lo < hi:
            mid = (lo + hi) // 2
            if self._below(n, mid, self.count) >= pos + 1:
                hi = mid
            else:
                lo = mid + 1
        return lo - 1

    def _from(self, start):
        """the values from position `start` on"""
        if start >= self.count:
            return
        for t in range(self._t_of(start), self.span):
            g = t ^ (t >> 1)
            if g < self.count:
                yield self.rank + self.world * g

    def __iter__(self):
        return self._from(0)

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            start, stop, step = idx.indices(self.count)
            if step <= 0:
                raise ValueError("Gray-ordered shards are sliced forwards")
            out, pos = [], start
            for v in self._from(start):
                if pos >= stop:
                    break
                if (pos - start) % step == 0:
                    out.append(v)
                pos += 1
            return out
        idx = int(idx)
        if idx < 0:
            idx += self.count
        if not 0 <= idx < self.count:
            raise IndexError("slice position out of range")
        t = self._t_of(idx)
        return self.rank + self.world * (t ^ (t >> 1))

    def __eq__(self, other):
        try:
            return len(other) == self.count and all(a == b for a, b in zip(self, other))
        except TypeError:
            return NotImplemented

    def __repr__(self):
        return f"<Gray-ordered shard: {self.count} slices of rank {self.rank} / {self.world}>"


def boundary_source_sha16():
    """sha256 (first 16 hex digits) of the two files that ARE the drop-in boundary -- contraction.py (compilers,
    executors) and simulation.py (slice loop, entry points).  tests/golden/check_boundary.py stamps its record with it;
    tests/test_boundary_record.py fails when the record was made for other sources."""
    import hashlib
    import os
    h = hashlib.sha256()
    here = os.path.dirname(os.path.abspath(__file__))
    for name in ("contraction.py", "simulation.py"):
        with open(os.path.join(here, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def apply_slice(tensors, slicing_indices, config):
    """Leaf tensors with the sliced bonds fixed to `config` (one bit per bond, in the
    mapping's order).  `slicing_indices[bond] = [(tensor_id, dim_index), ...]` with
    dim indices of the UNSLICED tensors (reference simulation.py:60-65).

    The reference applies `select(dim, bit)` bond after bond with those indices
    (simulation.py:110-113), which goes stale when one tensor carries two sliced bonds in
    ascending dim order.  Here all of a tensor's selects are applied
    in one indexing operation on the unsliced tensor, which equals the reference wherever
    the reference is well defined and is correct where it is not."""
    per_tensor = {}
    for x, (bond, lst) in enumerate(slicing_indices.items()):
        for tid, dim in lst:
            per_tensor.setdefault(tid, {})[dim] = config[x]
    out = dict(tensors) if isinstance(tensors, dict) else list(tensors)
    for tid, sel in per_tensor.items():
        t = tensors[tid]
        index = tuple(sel.get(d, slice(None)) for d in range(t.dim()))
        out[tid] = t[index].contiguous()
    return out


def accumulate(acc, x):
    """acc += x through artn_axpy_c64 (reference simulation.py:114 `collect_tensor += ...`)."""
    N.require_gpu(acc, "accumulate")
    N.require_gpu(x, "accumulate")
    if acc.shape != x.shape or acc.dtype != x.dtype or acc.dtype not in (torch.complex64, torch.complex128):
        raise RuntimeError(f"accumulate needs equal-shape complex64 (or complex128) tensors, got {acc.shape} {acc.dtype} / {x.shape} {x.dtype}")
    if not acc.is_contiguous():
        raise RuntimeError("accumulator must be contiguous")
    x = x.contiguous()
    axpy = N.lib().artn_axpy_c64 if acc.dtype == torch.complex64 else N.lib().artn_axpy_c128
    with torch.cuda.device(acc.device):
        N.check(axpy(acc.data_ptr(), x.data_ptr(), acc.numel(), N.current_stream_ptr(acc.device)))
    return acc


SMALL_NUMEL = 1 << 16  # operands and result of a "small" step have at most this many elements


def split_scheme(scheme, shapes):
    """Separate the steps that only combine small, leaf-derived tensors from the rest.

    A circuit scheme is a few big steps on the state tensor plus hundreds of tiny ones that
    build the small operands of later big steps out of leaf tensors; none of the tiny ones
    depends on the state.  Returns (small, main, out_shapes): `small` = indices of un-chunked steps
    whose operands are leaves or results of other small steps and whose operands and result
    have at most SMALL_NUMEL elements; `main` = the remaining steps in scheme order (a valid
    scheme once the results of the small steps are supplied as tensors).  `shapes` maps tensor
    id -> leaf shape."""
    shapes = dict(shapes)
    tainted = set()      # ids written by a main step: what they hold is not a small tensor any more
    main_reads = set()   # ids a main step reads as its second operand: a later step that overwrites one
                         # must stay behind that main step (the executors' semantics are sequential)
    small, main = [], []

    def numel(sh):
        n = 1
        for e in sh:
            n *= e
        return n

    for n, step in enumerate(scheme):
        i, j = step[0]
        out_shape = None
        if (i not in tainted and j not in tainted and i not in main_reads and i in shapes and j in shapes
                and (len(step) == 2 or (len(step[2][0]) <= 1 and len(step[2][1]) <= 1))):
            out_shape = _small_step_shape(step, shapes[i], shapes[j])
        if out_shape is not None and max(numel(shapes[i]), numel(shapes[j]), numel(out_shape)) <= SMALL_NUMEL:
            small.append(n)
            shapes[i] = out_shape
        else:
            main.append(step)
            tainted.add(i)
            main_reads.add(j)
            shapes.pop(i, None)
    return small, main, shapes


def _small_step_shape(step, si, sj):
    """Result shape of one dense step or one un-chunked sparse step (the branches of
    contraction._sparse_step / reference contraction.py:176-191), None if the equation does not
    match the operand ranks."""
    sparse5 = len(step) > 3
    bi, bj = step[2] if len(step) > 2 else ([], [])
    both = sparse5 and len(bi) == 1 and len(bj) == 1
    if both:  # branch (B): one row gather per operand, then the batched contraction
        si, sj = (len(bi[0]),) + tuple(si[1:]), (len(bj[0]),) + tuple(sj[1:])
    la, lb, lo = _labels(step[1])
    if len(la) != len(si) or len(lb) != len(sj):
        return None
    ext = dict(zip(la, si))
    ext.update(zip(lb, sj))
    out = tuple(ext[x] for x in lo)
    if sparse5 and not both:  # branch (C): reshape merges the batch labels, optional row select
        n = 1
        for e in out:
            n *= e
        rs = list(step[3])
        known = 1
        for e in rs:
            if e != -1:
                known *= e
        if known == 0 or n % known:
            return None
        out = tuple(n // known if e == -1 else e for e in rs)
        if len(bi) == 1:
            out = (len(bi[0]),) + out[1:]
    return out


class SliceRunner:
    """Runs slices of one sliced contraction on one device and accumulates them.

    Every slice executes the same scheme on leaf tensors of the same shapes; only the contents
    of the sliced leaves differ, and a small intermediate tensor depends only on the sliced
    bonds that touch the leaves below it.  The runner therefore evaluates the small steps
    itself (split_scheme) and keeps each result together with the values of the bonds it
    depends on: a slice recomputes only the small steps whose bonds changed since the previous
    slice (with Gray-ordered slices: the two root paths of ONE bond instead of ~340 launches of
    ~10 us each on n53 m14), then hands the remaining big steps to the executor.

    graph=True instead captures one whole slice (all steps) into a HIP graph and replays it:
    no per-launch host work at all, for hosts that cannot keep the GPU busy.
    """

    def __init__(self, tensors, scheme, slicing_indices, out_shape, sparse=False, dtype=torch.complex64,
                 device="cuda", graph=False, reuse_small=True):
        self._setup(tensors, scheme, slicing_indices, out_shape, sparse, dtype, device, graph, reuse_small, None, None)

    @classmethod
    def _with_seams(cls, tensors, scheme, slicing_indices, out_shape, sparse, dtype, device, execute, accumulate):
        """Test seam (tests/test_distributed.py): a runner whose executor / accumulator are injected, so
        the sharding and reduction logic can run on CPU boxes.  Never used by the product path."""
        self = cls.__new__(cls)
        self._setup(tensors, scheme, slicing_indices, out_shape, sparse, dtype, device, False, False, execute, accumulate)
        return self

    def _setup(self, tensors, scheme, slicing_indices, out_shape, sparse, dtype, device, graph, reuse_small,
               _execute, _accumulate):
        self.execute = _execute or (tensor_contraction_sparse if sparse else tensor_contraction)
        self.add = _accumulate or accumulate
        self.scheme = scheme
        self.device = torch.device(device)
        items = tensors.items() if isinstance(tensors, dict) else enumerate(tensors)
        self.leaves = {k: (t.to(dtype).to(device).contiguous() if isinstance(t, torch.Tensor) else t) for k, t in items}
        self.slicing_indices = slicing_indices or {}
        self.n_bonds = len(self.slicing_indices)
        self.collect = torch.zeros(tuple(out_shape), dtype=dtype, device=device)
        seams = _execute is not None or _accumulate is not None
        # dense schemes (complex64, and complex128 on artn_k_bits128): `collect += result` rides in the store phase of the slice's last launch where that launch is
        # a state-streaming step or pair (tensor_contraction(accumulate_into=...)); ARTN_NO_ACC=1 keeps the separate add
        self._fused_add = (not seams and not sparse and dtype in (torch.complex64, torch.complex128) and self.device.type == "cuda"
                           and __import__("os").environ.get("ARTN_NO_ACC", "0") in ("", "0"))
        on_gpu = not seams and self.device.type == "cuda" and self.n_bonds > 0
        self.use_graph = graph is True and on_gpu
        self._selects = {}    # tensor id -> {dim: position of the bond in the slice configuration}
        for x, (bond, lst) in enumerate(self.slicing_indices.items()):
            for tid, dim in lst:
                self._selects.setdefault(tid, {})[dim] = x
        self._static = None
        self._graph = None
        self._eager_done = 0
        # reuse of small intermediates across slices
        self.reuse_small = bool(reuse_small) and on_gpu and not self.use_graph
        self.small_steps_run = 0
        if self.reuse_small:
            shapes = {}
            for k, t in self.leaves.items():
                if isinstance(t, torch.Tensor):
                    sel = self._selects.get(k, {})
                    shapes[k] = tuple(e for d, e in enumerate(t.shape) if d not in sel)
            small, main, _ = split_scheme(scheme, shapes)
            self._small, self._main = small, main
            rel = {tid: frozenset(sel.values()) for tid, sel in self._selects.items()}
            self._step_rel = {}
            for n in small:
                i, j = scheme[n][0]
                rel[i] = rel.get(i, frozenset()) | rel.get(j, frozenset())
                self._step_rel[n] = tuple(sorted(rel[i]))
            self._leaf_rel = {tid: tuple(sorted(sel.values())) for tid, sel in self._selects.items()}
            self._memo = {}
            self._last_id = scheme[-1][0][0]
            # small steps that are one plain einsum on small tensors can be recomputed in batches, each batch ONE
            # launch of the small-step program (artn_program_*): recs[n] = labels and shapes, the same for every slice
            self._recs, self._progs = {}, {}
            self.program_launches = 0
            self._dtype = dtype
            if dtype in (torch.complex64, torch.complex128) and __import__("os").environ.get("ARTN_NO_PROGRAM", "0") in ("", "0"):
                cur_shapes = dict(shapes)
                numel = lambda sh: int(np.prod(sh, dtype=np.int64)) if len(sh) else 1
                for n in small:
                    step = scheme[n]
                    i, j = step[0]
                    out_shape = _small_step_shape(step, cur_shapes[i], cur_shapes[j])
                    if _C._is_plain_step(step) and 0 < max(numel(cur_shapes[i]), numel(cur_shapes[j]), numel(out_shape)) <= _C.PROGRAM_MAX_NUMEL:
                        la, lb, lo = _labels(step[1])
                        self._recs[n] = (la, lb, lo, tuple(cur_shapes[i]), tuple(cur_shapes[j]), tuple(out_shape))
                    cur_shapes[i] = out_shape

    def _index(self, tid, cfg):
        sel, t = self._selects[tid], self.leaves[tid]
        return tuple(cfg[sel[d]] if d in sel else slice(None) for d in range(t.dim()))

    def _load(self, cfg):
        for tid in self._selects:
            self._static[tid].copy_(self.leaves[tid][self._index(tid, cfg)])

    def _one(self, inputs):
        if self._fused_add:   # dense schemes: `collect += result` in the store phase of the slice's last launch
            self.execute(inputs, self.scheme, accumulate_into=self.collect)
            return
        res = self.execute(inputs, self.scheme)
        self.add(self.collect, res.reshape(self.collect.shape))

    def _one_reusing(self, cfg):
        memo = self._memo
        cur = dict(self.leaves)
        for tid, pos in self._leaf_rel.items():
            key = tuple(cfg[x] for x in pos)
            hit = memo.get(("leaf", tid))
            if hit is None or hit[0] != key:
                hit = memo[("leaf", tid)] = (key, self.leaves[tid][self._index(tid, cfg)].contiguous())
            cur[tid] = hit[1]
        # stale small steps are recomputed; runs of plain ones go into one program launch each (a stale step only feeds
        # stale steps -- its bonds are a subset of theirs -- and every tensor version is read once, so a batch can be
        # deferred while the walk goes on: what it reads from outside is captured when the step joins the batch)
        batch, batch_ext, produced = [], {}, set()

        def run_one(n, src):
            step = self.scheme[n]
            i, j = step[0]
            if len(step) == 2:
                return contract(step[1], src[i], src[j])
            scratch = {i: src[i], j: src[j]}   # sparse step: the executor's own branch logic on a scratch pair
            _sparse_step(scratch, step)
            return scratch[i]

        def flush():
            if not batch:
                return
            prog = None
            if len(batch) >= _C.PROGRAM_MIN_STEPS:
                pk = tuple(batch)
                if pk not in self._progs:
                    if len(self._progs) > 256:
                        self._progs.clear()
                    self._progs[pk] = _C._build_program(self.scheme, list(batch), self._recs, None, self._dtype)
                prog = self._progs[pk]
            if prog is not None:
                with torch.cuda.device(self.device):
                    ws = _C._run_program(prog, batch_ext, self._dtype, self.device, N.current_stream_ptr(self.device))
                self.program_launches += 1
                for n in batch:
                    off, shape = prog.step_out[n]
                    memo[n] = (tuple(cfg[x] for x in self._step_rel[n]), _C._ws_view(ws, off, shape, self._dtype))
            else:
                src = dict(batch_ext)
                for n in batch:
                    val = run_one(n, src)
                    src[self.scheme[n][0][0]] = val
                    memo[n] = (tuple(cfg[x] for x in self._step_rel[n]), val)
            for n in batch:
                cur[self.scheme[n][0][0]] = memo[n][1]
            self.small_steps_run += len(batch)
            batch.clear()
            batch_ext.clear()
            produced.clear()

        for n in self._small:
            i, j = self.scheme[n][0]
            key = tuple(cfg[x] for x in self._step_rel[n])
            hit = memo.get(n)
            if hit is not None and hit[0] == key:
                cur[i] = hit[1]
                continue
            if n in self._recs:
                for t in (i, j):
                    if t not in produced and t not in batch_ext:
                        batch_ext[t] = cur[t]
                produced.add(i)
                batch.append(n)
                continue
            flush()
            val = run_one(n, cur)
            memo[n] = (key, val)
            self.small_steps_run += 1
            cur[i] = val
        flush()
        if self._main and self._fused_add:
            self.execute(cur, self._main, accumulate_into=self.collect)
            return
        res = self.execute(cur, self._main) if self._main else cur[self._last_id]
        self.add(self.collect, res.reshape(self.collect.shape))

    def run(self, slices):
        """Contract the given slice numbers and add them to `self.collect` (returned).  The gather
        kernels' out-of-range flag is read once, after the last slice (one sync per call, not per
        slice), and raises RuntimeError -- the reference aborts on the first bad step."""
        prev = getattr(_C._defer, "flag_check", False)
        _C._defer.flag_check = True
        try:
            self._run(slices)
        finally:
            _C._defer.flag_check = prev
        if _C._flags_used() and not prev:
            _C.check_gather_flag("SliceRunner.run")
        return self.collect

    def _run(self, slices):
        for s in slices:
            cfg = slice_assignments(self.n_bonds, s)
            if self.reuse_small:
                self._one_reusing(cfg)
                continue
            if not self.use_graph:
                sliced = apply_slice(self.leaves, self.slicing_indices, cfg) if self.n_bonds else dict(self.leaves)
                self._one(sliced)
                continue
            if self._static is None:
                self._static = dict(self.leaves)
                for tid in self._selects:
                    self._static[tid] = torch.empty_like(self.leaves[tid][self._index(tid, cfg)],
                                                         memory_format=torch.contiguous_format)
            self._load(cfg)
            if self._eager_done < 1:        # first slice: eager, fills every host-side cache
                self._one(dict(self._static))
                self._eager_done += 1
                continue
            if self._graph is None:         # second slice: capture (nothing executes), then replay
                torch.cuda.synchronize(self.device)
                torch.cuda.empty_cache()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.device(self.device), torch.cuda.graph(g):
                    self._one(dict(self._static))
                self._graph = g
            self._graph.replay()
        return self.collect


def sliced_contraction(tensors, scheme, slicing_indices, out_shape, sparse=False, permute_dims=None,
                       dtype=torch.complex64, device="cuda", group=None, slices=None, reduce="all",
                       graph=False, reuse_small=True, runner=None, checkpoint=None, checkpoint_every=256):
    """The slice loop (reference simulation.py:101-116) on one rank of `group`.

    tensors         leaf tensors (dict or list) already on `device` or movable to it
    slices          explicit slice numbers for this rank (default: round-robin shard)
    reduce          "all" -> every rank returns the full sum (all_reduce);
                    "root" -> only rank 0 does (reduce to 0); None -> local partial sum
    reuse_small     keep small intermediate tensors across slices and recompute only those whose
                    sliced bonds changed (see SliceRunner); the default shard is then visited
                    in Gray-code order
    graph           True: replay whole slices from a captured HIP graph instead
    runner          a SliceRunner to reuse across calls (its accumulator is zeroed first)
    checkpoint      path prefix: this rank's partial sum and the slices it covers are written to
                    `<checkpoint>.rank<r>of<w>.pt` every `checkpoint_every` slices (atomically: write + rename) and when the
                    shard is done; a later call with the same plan, shard and path resumes after the last saved slice (a
                    16 384-slice n53 run is minutes per GPU, a 2^29-slice one is not a single sitting).  The reference has no
                    such thing (simulation.py:107-114 is one uninterrupted loop).
    """
    if runner is None:
        runner = SliceRunner(tensors, scheme, slicing_indices, out_shape, sparse=sparse, dtype=dtype, device=device,
                             graph=graph, reuse_small=reuse_small)
    else:
        runner.collect.zero_()
    return _shard_and_reduce(runner, permute_dims, group, slices, reduce, checkpoint, checkpoint_every)


def _checkpoint_path(prefix, rank, world):
    return f"{prefix}.rank{rank}of{world}.pt"


def _slices_array(my_slices):
    """the shard's slice numbers as one int64 array (8 bytes per slice instead of a list of Python ints)"""
    if isinstance(my_slices, np.ndarray) and my_slices.dtype == np.int64:
        return my_slices
    return np.fromiter((int(x) for x in my_slices), dtype=np.int64, count=len(my_slices))


def _checkpoint_head(my_slices, done):
    """the last (up to) eight slices a checkpoint covers: O(1) per save, whatever `done` is"""
    done = int(done)
    return [int(x) for x in my_slices[max(0, done - 8):done]]


def run_digest(runner, my_slices):
    """What a checkpoint must match besides the plan: the VALUES of the leaves (same circuit structure, other gate
    parameters: same plan fingerprint, different sums), their dtype, the accumulator's shape and dtype, and the whole
    slice list of this shard (sha256; the leaves of a circuit are a few KB each)."""
    import hashlib
    h = hashlib.sha256()
    for k in sorted(runner.leaves, key=repr):
        t = runner.leaves[k]
        h.update(repr(k).encode())
        if isinstance(t, torch.Tensor):
            h.update(repr((str(t.dtype), tuple(t.shape))).encode())
            h.update(np.ascontiguousarray(torch.view_as_real(t).cpu().numpy() if t.is_complex() else t.cpu().numpy()).tobytes())
    h.update(repr((str(runner.collect.dtype), tuple(runner.collect.shape))).encode())
    h.update(_slices_array(my_slices).tobytes())
    return h.hexdigest()


def load_checkpoint(path, fingerprint, my_slices, digest=None, collect=None):
    """(slices done, partial sum as a CPU tensor) of a checkpoint written for THIS plan, THESE leaf values and THIS shard, or
    (0, None).  A file of another plan (fingerprint), of other leaves / dtype / output shape / slice list (digest) is refused:
    resuming it would sum slices of a different network, or the same slices twice -- and a FINISHED file of another run
    would be returned as this run's result without contracting anything.  The prefix must be unique per run."""
    import os
    if not os.path.exists(path):
        return 0, None
    ck = torch.load(path, map_location="cpu", weights_only=True)
    if ck.get("fingerprint") != fingerprint:
        raise RuntimeError(f"{path} was written for another plan (scheme / sliced bonds / output order differ): not resumed")
    done = int(ck["done"])
    if ck.get("n_slices") != len(my_slices) or ck.get("head") != _checkpoint_head(my_slices, done):
        raise RuntimeError(f"{path} was written for another shard of the slices: not resumed")
    if digest is not None and ck.get("digest") != digest:
        raise RuntimeError(f"{path} was written for other leaf tensors, another dtype, output shape or slice list: not resumed")
    partial = ck["partial"]
    if collect is not None and (partial.dtype != collect.dtype or tuple(partial.shape) != tuple(collect.shape)):
        raise RuntimeError(f"{path} holds a {partial.dtype} partial sum of shape {tuple(partial.shape)}, this run accumulates "
                           f"{collect.dtype} {tuple(collect.shape)}: not resumed")
    return done, partial


def save_checkpoint(path, fingerprint, my_slices, done, collect, digest=None):
    """Atomic: the previous checkpoint stays valid until the new one is complete on disk."""
    import os
    tmp = path + ".tmp"
    torch.save({"fingerprint": fingerprint, "n_slices": len(my_slices), "done": int(done), "digest": digest,
                "head": _checkpoint_head(my_slices, done), "partial": collect.detach().cpu()}, tmp)
    os.replace(tmp, path)


def _all_ranks_ok(err, group, what):
    """A failure on ONE rank (a foreign checkpoint, a slice that raised) must not leave the others waiting in the
    collective that follows: every rank reports, every rank raises."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    all_ = [None] * world
    dist.all_gather_object(all_, None if err is None else f"{type(err).__name__}: {err}", group=group)
    bad = [(r, e) for r, e in enumerate(all_) if e is not None]
    if bad:
        if err is not None:
            raise err
        raise RuntimeError(f"{what} failed on rank {bad[0][0]}: {bad[0][1]}")


def plan_fingerprint(scheme, slicing_indices, permute_dims=None):
    """sha256 over what every rank of a sliced contraction must agree on: the steps (edges, equations, index
    tensors), the sliced bonds in order with their (tensor, dim) lists, and the output permutation.  The
    reference's planner depends on PYTHONHASHSEED (bond labels pass through set(): three seeds give three
    different trees), so ranks that each plan for themselves may hold DIFFERENT slicings of the same
    network; summing their slices would be silently wrong."""
    import hashlib
    h = hashlib.sha256()
    for step in scheme:
        h.update(repr((tuple(step[0]), step[1] if isinstance(step[1], str) else tuple(map(tuple, step[1])))).encode())
        if len(step) > 2:
            for lst in step[2]:
                for idx in lst:
                    h.update(np.ascontiguousarray(torch.as_tensor(idx).cpu().numpy()).tobytes())
            h.update(repr(tuple(step[3]) if len(step) > 3 and step[3] is not None else None).encode())
    h.update(repr([(str(b), [tuple(x) for x in v]) for b, v in (slicing_indices or {}).items()]).encode())
    h.update(repr(None if permute_dims is None else [int(x) for x in permute_dims]).encode())
    return h.hexdigest()


def _check_same_plan(runner, permute_dims, group):
    """Every rank of the group must run the same scheme on the same slicing (one all_gather of a 64-byte digest,
    once per runner and group)."""
    import torch.distributed as dist
    seen = getattr(runner, "_plan_checked", None)
    if seen is not None and seen == id(group):
        return
    mine = plan_fingerprint(runner.scheme, runner.slicing_indices, permute_dims)
    world = dist.get_world_size(group)
    all_ = [None] * world
    dist.all_gather_object(all_, mine, group=group)
    if any(x != mine for x in all_):
        raise RuntimeError("the ranks of this sliced contraction hold different plans (scheme / sliced bonds / output order): "
                           "the reference's planner depends on PYTHONHASHSEED -- plan on one rank and broadcast the plan "
                           "(tensor_network_contraction does), or pin PYTHONHASHSEED identically on every rank")
    runner._plan_checked = id(group)


def _shard_and_reduce(runner, permute_dims=None, group=None, slices=None, reduce="all", checkpoint=None, checkpoint_every=256):
    """Sharding + the single reduction of sliced_contraction around a ready SliceRunner (also the entry
    of the CPU tests, which hand in a runner built by SliceRunner._with_seams)."""
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized() and (group is not None or dist.get_world_size() > 1)
    rank = dist.get_rank(group) if distributed else 0
    world = dist.get_world_size(group) if distributed else 1
    if distributed and reduce is not None:
        _check_same_plan(runner, permute_dims, group)
    if slices is None:
        slices = rank_slices(2 ** runner.n_bonds, rank, world, gray=runner.reuse_small)
    guarded = distributed and reduce is not None
    if checkpoint is None:
        # (a slice that raises on ONE rank must not leave the others waiting in the reduction: every rank reports first)
        err, collect = None, None
        try:
            collect = runner.run(slices)
        except Exception as e:
            if not guarded:
                raise
            err = e
        if guarded:
            _all_ranks_ok(err, group, "the slice loop")
    else:   # resumable: the shard in pieces of `checkpoint_every` slices, the partial sum saved after each
        fp = plan_fingerprint(runner.scheme, runner.slicing_indices, permute_dims)
        path = _checkpoint_path(checkpoint, rank, world)
        err, done, partial, dg = None, 0, None, None
        try:
            # (inside the guarded block: shards differ by one slice, so ONE rank may be the only one over the limit)
            if len(slices) > (1 << 26):
                raise RuntimeError(f"checkpointing keeps the shard's slice list ({len(slices)} slices): pass `slices=` pieces of at most 2^26")
            slices = np.fromiter((int(x) for x in slices), dtype=np.int64, count=len(slices))
            dg = run_digest(runner, slices)
            done, partial = load_checkpoint(path, fp, slices, dg, runner.collect)
        except Exception as e:   # (reported to every rank below: nobody enters the reduction alone)
            err = e
        if guarded:
            _all_ranks_ok(err, group, "loading the checkpoint")
        elif err is not None:
            raise err
        if partial is not None:
            runner.collect.copy_(partial.to(runner.collect.device))
        every = max(1, int(checkpoint_every))
        try:
            while done < len(slices):
                runner.run(slices[done:done + every].tolist())
                done = min(done + every, len(slices))
                save_checkpoint(path, fp, slices, done, runner.collect, dg)
        except Exception as e:
            err = e
        if guarded:
            _all_ranks_ok(err, group, "the slice loop")
        elif err is not None:
            raise err
        collect = runner.collect
    if distributed and reduce is not None:
        buf = torch.view_as_real(collect)
        if reduce == "all":
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        else:
            # (torch wants the GLOBAL rank of the destination: rank 0 of a sub-group is not global rank 0)
            dist.reduce(buf, dst=dist.get_global_rank(group, 0) if group is not None else 0, op=dist.ReduceOp.SUM, group=group)
    if permute_dims is not None and len(permute_dims) > 0:
        collect = collect.permute(tuple(permute_dims))
    return collect


# ----------------------------------------------------------------------------------------
# output-qubit partitioning of an unsliced dense contraction (build-side extension)
# ----------------------------------------------------------------------------------------
def _step_cost(labels, ext):
    n = 8.0
    for x in labels:
        n *= ext[x]
    return n


def partition_output(scheme, leaf_shapes, n_fix):
    """Fix `n_fix` dims of the FINAL result of a dense scheme to constants, at the leaves.

    The reference can only shard a contraction over sliced inner bonds, which for a dense
    full-amplitude output ends in a sum of whole output tensors (simulation.py:107-114; SURVEY 8e: an
    8 GiB all-reduce for n30).  An output label, though, is a dangling bond of one leaf tensor that
    every step merely carries along: selecting one value of it AT THE LEAF gives the same scheme with
    that label dropped everywhere, whose result is one slab of the full output -- 2^n_fix independent
    contractions with disjoint results and no exchange at all.

    Returns (new_scheme, selects, fixed_dims): `new_scheme` in the format of `scheme` (einsum strings
    or label tuples) with the chosen labels removed from every step; `selects` = [(leaf id, leaf dim)]
    in the order of `fixed_dims`, the dims of the original final result they fix (value v of select q
    <-> index v along fixed_dims[q]).  Dims are chosen greedily by how much work fixing them removes
    (labels that join the state early are carried by every big step)."""
    from .contraction import _labels
    prov = {k: [frozenset([(k, d)]) for d in range(len(sh))] for k, sh in leaf_shapes.items()}
    ext0 = {(k, d): e for k, sh in leaf_shapes.items() for d, e in enumerate(sh)}
    steps = []
    for (i, j), eq in (st[:2] for st in scheme):
        la, lb, lo = _labels(eq)
        org = {}
        for x, o in zip(la, prov[i]):
            org[x] = o
        for x, o in zip(lb, prov[j]):
            org[x] = org[x] | o if x in org else o
        steps.append((i, j, la, lb, lo, org))
        prov[i] = [org[x] for x in lo]
    final_id = scheme[-1][0][0]
    final = prov[final_id]

    def total_cost(fixed):
        c = 0.0
        for i, j, la, lb, lo, org in steps:
            n = 8.0
            for x in dict.fromkeys(la + lb):
                if org[x] & fixed:
                    continue
                n *= ext0[next(iter(org[x]))]
            c += n
        return c

    fixed, fixed_dims = frozenset(), []
    for _ in range(n_fix):
        best = None
        for d, o in enumerate(final):
            if d in fixed_dims or len(o) != 1:
                continue
            c = total_cost(fixed | o)
            if best is None or c < best[0] or (c == best[0] and d < best[1]):
                best = (c, d, o)
        if best is None:
            raise RuntimeError("not enough single-leaf output labels to partition over")
        fixed = fixed | best[2]
        fixed_dims.append(best[1])
    selects = [next(iter(final[d])) for d in fixed_dims]
    new_scheme = []
    for (i, j, la, lb, lo, org), st in zip(steps, scheme):
        keep = lambda labs: tuple(x for x in labs if not (org[x] & fixed))
        ka, kb, ko = keep(la), keep(lb), keep(lo)
        eq = ("".join(ka) + "," + "".join(kb) + "->" + "".join(ko)) if isinstance(st[1], str) else (ka, kb, ko)
        new_scheme.append(((i, j), eq))
    return new_scheme, selects, fixed_dims


_partition_cache = _C._IdMemo(64)   # (scheme; n_fix, leaf shapes) -> (partitioned scheme, selects, fixed dims)


def partitioned_contraction(tensors, scheme, n_fix, part, device="cuda", dtype=torch.complex64):
    """Slab `part` (0 <= part < 2^n_fix) of the result of a dense scheme: the output dims chosen by
    partition_output fixed to the bits of `part` (MSB = first fixed dim).  Returns
    (slab, fixed_dims, values); slab has the result's remaining dims in order."""
    items = list(tensors.items()) if isinstance(tensors, dict) else list(enumerate(tensors))
    shapes = {k: tuple(t.shape) for k, t in items}
    extra = (n_fix, tuple(sorted(shapes.items(), key=lambda kv: repr(kv[0]))))
    hit = _partition_cache.find((scheme,), extra, (scheme,))
    if hit is _C._MISS:
        hit = _partition_cache.keep((scheme,), tuple(partition_output(scheme, shapes, n_fix)), extra, (scheme,))
    new_scheme, selects, fixed_dims = hit
    values = slice_assignments(n_fix, part)
    per_leaf = {}
    for (leaf, dim), v in zip(selects, values):
        per_leaf.setdefault(leaf, {})[dim] = v
    leaves = {}
    for k, t in items:
        t = t.to(dtype).to(device)
        if k in per_leaf:
            t = t[tuple(per_leaf[k].get(d, slice(None)) for d in range(t.dim()))].contiguous()
        leaves[k] = t
    return tensor_contraction(leaves, new_scheme), fixed_dims, values


def plan_output_slabs(tensor_bonds, bond_dims, final_qubits, n_slabs, candidates=None, planner=None, **plan_args):
    """Re-plan an unsliced dense (full-amplitude) network for `n_slabs` = 2^k ranks WITHOUT a collective (round 5).

    partition_output fixes k output labels at the leaves of the tree planned for the FULL network; here the REDUCED
    network -- the dangling bonds of k chosen output qubits removed from their leaves -- goes back to the reference's
    planner (order_finder.py:174-198), which is free to find a tree for it.  On Sycamore n30 m14 the slabs then cost
    0.85 / 0.88 / 0.97 x the unsliced FLOP IN TOTAL at N = 2 / 4 / 8 (1.36 / 2.13 / 3.78 x with the one tree): the 8 ranks of
    a node together execute what one GPU executes alone.  Which qubits are fixed matters (0.97 x ... 11 x at N = 8), a
    plan of n30 takes 12 s: every run of k neighbouring final qubits is tried unless `candidates` (tuples of positions
    in sorted(final_qubits)) says otherwise.  Returns dict(fixed=[(leaf id, dim, qubit)...], tensor_bonds, final_qubits,
    ctree, log10_tc, candidates_tried).  Slab r: fixed[j] takes bit j of r."""
    import copy
    import math
    R = _planner(planner)
    k = int(round(math.log2(n_slabs)))
    if (1 << k) != n_slabs or k < 1:
        raise RuntimeError("n_slabs must be a power of two >= 2")
    fq = sorted(final_qubits)
    count = {}
    for tid, bonds in tensor_bonds.items():
        for b in bonds:
            count[b] = count.get(b, 0) + 1
    dangling = {}
    for q, tid in enumerate(fq):
        mine = [b for b in tensor_bonds[tid] if count[b] == 1]
        if len(mine) != 1:
            raise RuntimeError(f"final qubit {q} (tensor {tid}) has {len(mine)} dangling bonds: not a full-amplitude network")
        dangling[q] = mine[0]
    if candidates is None:
        candidates = [tuple(range(q, q + k)) for q in range(0, len(fq) - k + 1)]
    best, tried = None, []
    for cand in candidates:
        tb = {t: list(b) for t, b in tensor_bonds.items()}
        fixed = [(tid, tb[tid].index(dangling[q]), q) for q, tid in ((q, fq[q]) for q in cand)]
        for tid, dim, q in sorted(fixed, key=lambda x: -x[1]):
            tb[tid].pop(dim)
        fq2 = [t for q, t in enumerate(fq) if q not in cand]
        bd = {b: bond_dims[b] for b in set().union(*tb.values())}
        order, slicing_bonds, ctree = R.find_order(copy.deepcopy(tb), copy.deepcopy(bd), set(fq2), 0, 1, **plan_args)
        tc = float(ctree.tree_complexity()[0])
        tried.append((tuple(cand), tc, len(slicing_bonds)))
        if len(slicing_bonds) == 0 and (best is None or tc < best["log10_tc"]):
            best = dict(fixed=fixed, tensor_bonds=tb, final_qubits=fq2, ctree=ctree, log10_tc=tc, dangling=dangling)
    if best is None:
        raise RuntimeError("the planner sliced every reduced network: raise sc_target")
    best["candidates_tried"] = tried
    return best


def slab_contraction(tensors, scheme, fixed, part, device="cuda", dtype=torch.complex64):
    """Slab `part` of a full-amplitude contraction planned by plan_output_slabs (or loaded from a committed plan such as
    tests/golden/n30_dense_part8.npz): the leaves with fixed[j] = (leaf id, dim, qubit) set to bit j of `part`, contracted
    by `scheme` (the reduced network's own scheme).  Returns the raw slab (dims in the scheme's output order)."""
    items = list(tensors.items()) if isinstance(tensors, dict) else list(enumerate(tensors))
    per_leaf = {}
    for j, (leaf, dim, _q) in enumerate(fixed):
        per_leaf.setdefault(leaf, {})[int(dim)] = (int(part) >> j) & 1
    leaves = {}
    for k, t in items:
        t = t.to(dtype).to(device)
        if k in per_leaf:
            t = t[tuple(per_leaf[k].get(d, slice(None)) for d in range(t.dim()))].contiguous()
        leaves[k] = t
    return tensor_contraction(leaves, scheme)


def _planner(planner=None):
    """The planning front end (circuit parser, simplifier, order finder, contraction tree): the
    reference package itself -- `north_star` keeps it untouched -- or any module with the same names."""
    if planner is not None:
        return planner
    try:
        import artensor
    except ImportError as e:
        raise RuntimeError("planning stays with the reference's front end: install Fanerst/artensor next to "
                           "artensor_amd, or pass planner=<module with the same API>") from e
    return artensor


def _bond_owners(tensor_bonds):
    owners = {}
    for tid, bonds in (tensor_bonds.items() if isinstance(tensor_bonds, dict) else enumerate(tensor_bonds)):
        for b in bonds:
            owners.setdefault(b, []).append(tid)
    return owners


def _slicing_indices(slicing_bonds, tensor_bonds, tensors):
    """bond -> [(tensor id, dim)], dims of the ACTUAL leaf tensors.  The reference takes the bond's
    position in tensor_bonds[tid] (simulation.py:60-65, :176-178); in the sparse pattern a final-qubit
    leaf carries a leading batch dim that tensor_bonds does not list, so its select() would hit the
    batch dim there -- the offset is added here."""
    owners = _bond_owners(tensor_bonds)
    out = {}
    for bond in slicing_bonds:
        out[bond] = [(tid, tensor_bonds[tid].index(bond) + tensors[tid].dim() - len(tensor_bonds[tid]))
                     for tid in owners.get(bond, [])]
    return out


def _output_permutation(output_bonds, tensor_bonds, final_qubits, sparse):
    """Dims of the raw result -> final-qubit order (reference simulation.py:68-77, :186-195)."""
    if len(output_bonds) == 0:
        return None
    owners = _bond_owners(tensor_bonds)
    fq = list(final_qubits)
    where = []
    for b in output_bonds:
        tids = owners[b]
        if len(tids) != 1 or tids[0] not in fq:
            raise RuntimeError(f"output bond {b!r} is not a dangling bond of a final-qubit tensor")
        where.append(fq.index(tids[0]))
    perm = [int(x) for x in np.argsort(where)]
    return [0] + [d + 1 for d in perm] if sparse else perm


class TensorNetworkSimulation:
    """Execution half of the reference class of the same name (simulation.py:33-148).

    Planning (`prepare_contraction`, simulation.py:47-77) stays with the reference's
    planner; this class is built from its products -- either a planned reference object
    (`from_planned(sim)`, duck-typed) or a saved case (`from_case(case)`) -- and runs
    `.contraction(...)` with the reference's signature on the MI355X engine."""

    def __init__(self, tensors, scheme, slicing_indices, output_bonds, pattern="normal",
                 bitstrings_sorted=None, permute_dims=None):
        self.tensors = tensors
        self.scheme = scheme
        self.slicing_indices = slicing_indices or {}
        self.output_bonds = list(output_bonds)
        self.pattern = pattern
        self.bitstrings_sorted = bitstrings_sorted
        if permute_dims is not None:
            self.permute_dims = permute_dims
        self.tensor_contraction_func = tensor_contraction_sparse if pattern == "sparse" else tensor_contraction

    @classmethod
    def from_planned(cls, sim):
        """`sim`: a reference TensorNetworkSimulation after prepare_contraction().  Slicing indices are re-taken
        on the ACTUAL leaf tensors where the object carries `tensor_bonds` (a sparse final-qubit leaf has a leading
        batch dim that the reference's indices, simulation.py:60-65, do not count: `_slicing_indices`) -- the same
        correction prepare_contraction() and tensor_network_contraction() apply, so the three constructors agree."""
        slicing = sim.slicing_indices
        bonds = getattr(sim, "tensor_bonds", None)
        if slicing and bonds is not None:
            slicing = _slicing_indices(list(slicing.keys()), bonds, sim.tensors)
        return cls(sim.tensors, sim.scheme, slicing, sim.output_bonds, sim.pattern,
                   getattr(sim, "bitstrings_sorted", None), getattr(sim, "permute_dims", None))

    # ---- the reference's own constructors and planning calls, forwarded to its front end -------------
    @classmethod
    def _from_front(cls, front):
        self = cls(front.tensors, [], {}, [], front.pattern)
        self._front = front
        for name in ("tensor_bonds", "bond_dims", "final_qubits", "bitstrings", "max_bitstrings"):
            setattr(self, name, getattr(front, name))
        return self

    @classmethod
    def from_circuit_file(cls, circuit_filename, bitstrings=[], planner=None):
        """Reference simulation.py:120-133: the circuit is parsed and simplified by the front end."""
        return cls._from_front(_planner(planner).TensorNetworkSimulation.from_circuit_file(circuit_filename, bitstrings))

    @classmethod
    def from_tn_circuit(cls, circ, bitstrings=[], planner=None):
        """Reference simulation.py:135-148."""
        return cls._from_front(_planner(planner).TensorNetworkSimulation.from_tn_circuit(circ, bitstrings))

    def prepare_contraction(self, sc_target=30, **planner_args):
        """Reference simulation.py:47-77: the order finder runs in the front end (its keyword arguments
        pass through); the scheme is then compiled by this package (`update_scheme`), slicing indices
        are taken on the actual leaf tensors and the output permutation is recomputed from the tree."""
        front = getattr(self, "_front", None)
        if front is None:
            raise RuntimeError("prepare_contraction needs an object built by from_circuit_file / from_tn_circuit")
        front.prepare_contraction(sc_target=sc_target, **planner_args)
        self.ctree = front.ctree
        self.slicing_indices = _slicing_indices(list(front.slicing_indices.keys()), front.tensor_bonds, front.tensors)
        self.update_scheme(sc_target, self.bitstrings)
        perm = _output_permutation(self.output_bonds, front.tensor_bonds, front.final_qubits, self.pattern == "sparse")
        if perm is not None:
            self.permute_dims = perm
        return self

    def update_scheme(self, sc_target=30, bitstrings=[]):
        """Reference simulation.py:79-88, with this package's compilers."""
        from copy import deepcopy
        from .contraction import contraction_scheme, contraction_scheme_sparse
        if self.pattern == "normal":
            self.scheme, self.output_bonds = contraction_scheme(deepcopy(self.ctree))
        else:
            self.scheme, self.output_bonds, self.bitstrings_sorted = contraction_scheme_sparse(
                deepcopy(self.ctree), bitstrings, sc_target=sc_target)
            if len(self.bitstrings_sorted) > self.max_bitstrings:
                raise RuntimeError("more output rows than distinct bitstrings")
        self.output_bonds = list(self.output_bonds)

    @classmethod
    def from_case(cls, case):
        m = case.meta
        return cls(case.tensors, case.scheme, case.slicing_indices, m.get("output_bonds", []),
                   m.get("pattern", "normal"), m.get("bitstrings_sorted"), m.get("permute_dims"))

    def contraction(self, tensors=None, dtype=torch.complex64, device="cuda", group=None, reduce="all", checkpoint=None,
                    checkpoint_every=256):
        src = self.tensors if tensors is None else tensors
        if self.pattern == "sparse":
            shape = [len(self.bitstrings_sorted)] + [2] * len(self.output_bonds)
        else:
            shape = [2] * len(self.output_bonds)
        permute = getattr(self, "permute_dims", None) if len(self.output_bonds) > 0 else None
        return sliced_contraction(src, self.scheme, self.slicing_indices, shape, sparse=self.pattern == "sparse",
                                  permute_dims=permute, dtype=dtype, device=device, group=group, reduce=reduce,
                                  checkpoint=checkpoint, checkpoint_every=checkpoint_every)


def tensor_network_contraction(tensors, tensor_bonds, bond_dims, final_qubits, bitstrings=[], sc_target=31,
                               trial_num=8, alpha=0.0, dtype=torch.complex64, device="cuda", planner=None,
                               group=None, reduce="all"):
    """One-call API of the reference (simulation.py:151-213), same arguments and return value
    `(tensor, bitstrings)`.  Simplification and order finding run in the reference's front end
    (`planner`, default: the installed `artensor` package) with the reference's own settings
    (simulation.py:160-165: trials=trial_num, iters=50, betas 3..21 in 61 steps, start_seed 0); scheme
    compilation, the slice loop and every contraction run here.  `group`/`reduce`: slices are sharded
    over the ranks of a torch.distributed group with one reduction at the end (see sliced_contraction)."""
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized() and (group is not None or dist.get_world_size() > 1)

    def plan():
        from copy import deepcopy
        from .contraction import contraction_scheme, contraction_scheme_sparse
        P = _planner(planner)
        distinct = len(np.unique(bitstrings)) if len(bitstrings) else 0
        sparse = distinct > 0
        max_bitstrings = distinct if sparse else 1
        net = P.NumericalTensorNetwork(tensors, tensor_bonds, bond_dims, final_qubits)
        bonds_of, final_ids = net._simplify("sparse" if sparse else "normal")
        order, slicing_bonds, ctree = P.find_order(
            deepcopy(bonds_of), deepcopy(net.bond_dims), final_ids, 0, max_bitstrings, sc_target=sc_target,
            trials=trial_num, iters=50, betas=np.linspace(3.0, 21.0, 61), start_seed=0, alpha=alpha)
        leaves = {new: net.tensors[old] for new, old in enumerate(net.tensors.keys())}
        slicing = _slicing_indices(slicing_bonds, bonds_of, leaves)
        if sparse:
            scheme, output_bonds, sorted_bits = contraction_scheme_sparse(ctree, bitstrings, sc_target=sc_target)
            if len(sorted_bits) != max_bitstrings:
                raise RuntimeError("the compiled scheme does not produce one row per distinct bitstring")
            shape = [max_bitstrings] + [2] * len(output_bonds)
        else:
            scheme, output_bonds, sorted_bits = contraction_scheme(ctree) + (bitstrings,)
            shape = [2] * len(output_bonds)
        perm = _output_permutation(list(output_bonds), bonds_of, final_ids, sparse)
        return leaves, scheme, slicing, shape, sparse, perm, sorted_bits

    # The planner's trees, sliced bonds and output order depend on PYTHONHASHSEED, which torchrun
    # children do not share: ONE rank plans (simplified leaves included: their numbering follows the plan) and
    # the plan is broadcast, so every rank contracts slices of the same slicing.
    if distributed:
        # (a planner failure on rank 0 -- import error, an inconsistent scheme, out of memory while simplifying -- must not
        #  leave the other ranks waiting in the broadcast for ever: the exception travels in the plan's place and every
        #  rank raises.  With backend nccl the pickled plan moves through the current CUDA device: set it per rank first.)
        box = [None]
        if dist.get_rank(group) == 0:
            try:
                box = [("plan", plan())]
            except Exception as e:   # noqa: BLE001 -- re-raised on every rank below
                box = [("error", f"{type(e).__name__}: {e}")]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if box[0][0] == "error":
            raise RuntimeError(f"planning failed on rank 0: {box[0][1]}")
        leaves, scheme, slicing, shape, sparse, perm, sorted_bits = box[0][1]
    else:
        leaves, scheme, slicing, shape, sparse, perm, sorted_bits = plan()
    out = sliced_contraction(leaves, scheme, slicing, shape, sparse=sparse, permute_dims=perm, dtype=dtype,
                             device=device, group=group, reduce=reduce)
    return out, sorted_bits


def quantum_circuit_simulation(circuit_filename, bitstrings=[], sc_target=31, trial_num=8, alpha=0.0,
                               dtype=torch.complex64, device="cuda", planner=None):
    """Reference simulation.py:216-225: parse a qsim-format circuit with the front end's
    TensorNetworkCircuit, then tensor_network_contraction."""
    circ = _planner(planner).TensorNetworkCircuit(circuit_filename)
    tensors, tensor_bonds, bond_dims, final_qubits = circ.to_numerical_tn()
    return tensor_network_contraction(tensors, tensor_bonds, bond_dims, final_qubits, bitstrings, sc_target,
                                      trial_num, alpha, dtype, device, planner=planner)
