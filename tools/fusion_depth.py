#!/usr/bin/env python3
"""How deep could the state-streaming fusion go?  Host-only analysis of the committed schemes (VERDICT r03 missing #5:
"fusion beyond pairs ... not shown for n53 / random networks").

For every chain of consecutive big steps on the same first operand (what fusion_schedule pairs up) the tool asks whether
g = 2, 3, 4 consecutive steps could share ONE pass over HBM with a tile of 2^T elements (T = 12: two 32 KiB LDS regions per
workgroup, two workgroups per CU -- the shipped kernel; T = 13: 64 KiB regions, one workgroup per CU).  A group fits when

    * every contracted bit of every step of the group that already exists in the group's input tensor lies in the tile,
      together with the low RUN bits of the input (coalesced 16-byte lane loads in runs of 2^RUN elements) and those of
      the low RUN output bits that are old bits (new bits are produced inside the tile);
    * the tile never grows beyond 2^T elements after any stage (growth steps add more bits than they contract).

This is the planner's own criterion (artn_plan.h make_bits: K bits + run bits of A and C <= tile bits) extended to g stages;
for g = 2 it is checked against the shipped planner's answer (pair_info).  Bytes = 8 * (numel in + numel out) of the
big operand per pass (what a pass over HBM moves), summed over the scheme for the greedy grouping at each depth.

    python tools/fusion_depth.py [fixture.npz ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from artensor_amd.fixtures import load_case  # noqa: E402
from artensor_amd.contraction import _labels  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
BIG = 1 << 20   # steps whose first operand has at least this many elements run on the tiled kernels


def big_chains(case, sliced):
    """[(first-operand labels (fastest LAST), [(la, lb, lo), ...])]: maximal runs of consecutive big steps on one tensor."""
    shapes = {k: tuple(t.shape) for k, t in case.tensors.items()}
    if sliced and case.slicing_indices:
        for bond, lst in case.slicing_indices.items():
            pass
        per = {}
        for bond, lst in case.slicing_indices.items():
            for tid, dim in lst:
                per.setdefault(tid, set()).add(dim)
        # a sliced dim disappears from its leaf; labels follow the scheme's own equations, which are written for sliced leaves
        shapes = {k: tuple(e for d, e in enumerate(s) if d not in per.get(k, ())) for k, s in shapes.items()}
    chains, cur, cur_id = [], [], None
    for step in case.scheme:
        (i, j), eq = step[0], step[1]
        la, lb, lo = _labels(eq)
        ext = dict(zip(la, shapes[i]))
        ext.update(zip(lb, shapes[j]))
        if len(step) > 3 and step[3]:      # sparse 5-tuples: the reshape merges batch labels; shapes follow step[4] when present
            pass
        numel = 1
        for x in la:
            numel *= ext[x]
        plain = len(step) == 2 or (len(step) == 3 and len(step[2][0]) <= 1)
        pow2 = all((ext[x] & (ext[x] - 1)) == 0 for x in set(la) | set(lb))
        if numel >= BIG and plain and pow2:
            if cur_id != i:
                if cur:
                    chains.append(cur)
                cur, cur_id = [], i
            cur.append((tuple(la), tuple(lb), tuple(lo), dict(ext)))
        else:
            if cur and (i == cur_id or j == cur_id):
                chains.append(cur)
                cur, cur_id = [], None
        out_shape = tuple(ext[x] for x in lo)
        if len(step) > 3 and step[3]:
            import numpy as np
            n = int(np.prod(out_shape))
            tgt = [e for e in step[3]]
            known = int(np.prod([e for e in tgt if e != -1])) if tgt else 1
            out_shape = tuple(n // known if e == -1 else e for e in tgt)
        if len(step) > 3 and len(step[2][0]) == 1 and len(step[2][1]) != 1:
            out_shape = (len(step[2][0][0]),) + tuple(out_shape[1:])
        shapes[i] = out_shape
    if cur:
        chains.append(cur)
    return chains


def bits_of(labels, ext):
    """label list (slowest first) -> list of bit ids (label, bit) fastest FIRST"""
    out = []
    for x in reversed(labels):
        n = ext[x].bit_length() - 1
        out += [(x, b) for b in range(n)]
    return out


def group_fits(steps, T, run):
    """steps: consecutive (la, lb, lo, ext) on the same tensor.  Returns True when they can share one pass with a 2^T tile."""
    la0, _, _, ext0 = steps[0]
    in_bits = bits_of(la0, ext0)
    old = set(in_bits)
    need = set(in_bits[:run])                       # low bits of the input
    size_delta, max_delta = 0, 0
    alive = set(in_bits)
    for (la, lb, lo, ext) in steps:
        k_labels = [x for x in la if x in lb and x not in lo]
        n_labels = [x for x in lb if x not in la]
        k_bits = [b for x in k_labels for b in bits_of([x], ext)]
        n_bits = [b for x in n_labels for b in bits_of([x], ext)]
        need |= {b for b in k_bits if b in old}
        size_delta += len(n_bits) - len(k_bits)
        max_delta = max(max_delta, size_delta)
    out_bits = bits_of(steps[-1][2], steps[-1][3])
    need |= {b for b in out_bits[:run] if b in old}
    return len(need) + max(0, max_delta) <= T


def passes(chains, depth, T, run):
    """greedy grouping of every chain into groups of at most `depth` steps that fit; returns (n_passes, bytes moved)"""
    n, total = 0, 0.0
    for ch in chains:
        p = 0
        while p < len(ch):
            g = 1
            for want in range(depth, 1, -1):
                if p + want <= len(ch) and group_fits(ch[p:p + want], T, run):
                    g = want
                    break
            la, _, _, ext = ch[p]
            lo_last, ext_last = ch[p + g - 1][2], ch[p + g - 1][3]
            n_in = 1
            for x in la:
                n_in *= ext[x]
            n_out = 1
            for x in lo_last:
                n_out *= ext_last[x]
            total += 8.0 * (n_in + n_out)
            n += 1
            p += g
    return n, total


def main(names):
    print("| fixture | big steps (chains) | T | depth 1: passes / GB | depth 2 | depth 3 | depth 4 |")
    print("|---|---|---|---|---|---|---|")
    for name in names:
        case = load_case(os.path.join(GOLDEN, name))
        chains = big_chains(case, sliced=True)
        n_big = sum(len(c) for c in chains)
        for T, run in ((12, 4), (13, 4)):
            cells = []
            for depth in (1, 2, 3, 4):
                n, b = passes(chains, depth, T, run)
                cells.append(f"{n} / {b / 1e9:.1f}")
            print(f"| {name[:-4]} | {n_big} ({len(chains)}) | {T} | " + " | ".join(cells) + " |")


if __name__ == "__main__":
    main(sys.argv[1:] or ["n30_dense.npz", "n30_dense_sliced3.npz", "n53_m14_sliced.npz", "n53_m20_sliced.npz", "rand_D2_nv260_sliced.npz",
                          "n30_sparse10000.npz"])


def detail(name, T=12, run=4, depth=3):
    """Which groups the greedy grouping picks, with the tile bits each one needs."""
    case = load_case(os.path.join(GOLDEN, name))
    for ch in big_chains(case, sliced=True):
        p = 0
        while p < len(ch):
            g = 1
            for want in range(depth, 1, -1):
                if p + want <= len(ch) and group_fits(ch[p:p + want], T, run):
                    g = want
                    break
            ks = []
            for (la, lb, lo, ext) in ch[p:p + g]:
                ks.append((sum(ext[x].bit_length() - 1 for x in la if x in lb and x not in lo),
                           sum(ext[x].bit_length() - 1 for x in lb if x not in la)))
            la, _, _, ext = ch[p]
            rank = sum(ext[x].bit_length() - 1 for x in la)
            # bits needed at run = 4 and at run = 3
            need = {}
            for r in (4, 3):
                t = 0
                while not group_fits(ch[p:p + g], t, r):
                    t += 1
                need[r] = t
            print(f"  steps {p}..{p + g - 1}: rank {rank}, (k, n) = {ks}, tile bits needed: {need[4]} (128-byte runs) / {need[3]} (64-byte runs)")
            p += g
