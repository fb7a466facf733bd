#!/usr/bin/env python3
"""How deep can the state-streaming fusion go?  Host-only: the dense executor's own chain cut (contraction._cut_chain: a
dynamic programme over the planner's answers, artn_contract2_query / artn_contract3_query) on the committed schemes, with
three-step fusion off, on as shipped (triples need every contracted old bit + 128-byte runs inside one 2^12 tile, fragments
of at most 80 registers), and with the two limits relaxed (64-byte input runs, 96 fragment registers).

    python tools/fusion_depth.py            ->  the table of DESIGN.md section 4.1c (VERDICT r03 missing #5)

Per fixture: launches over tensors of 2^22+ elements and the bytes those launches move (8 B x (elements in + out))."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FIXTURES = ["n30_dense", "n30_dense_sliced3", "rand_D2_nv260_sliced", "rand_D4_nv100", "n53_m14_sliced", "n53_m20_sliced"]


def numel(s):
    r = 1
    for e in s:
        r *= e
    return r


def report():
    from artensor_amd import contraction as C
    from artensor_amd.fixtures import load_case
    for name in FIXTURES:
        case = load_case(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        shapes = {k: tuple(t.shape) for k, t in case.tensors.items()}
        if case.slicing_indices:   # one slice: the sliced dims are gone from the leaves
            per = {}
            for bond, lst in case.slicing_indices.items():
                for tid, dim in lst:
                    per.setdefault(tid, set()).add(dim)
            shapes = {k: tuple(e for d, e in enumerate(s) if d not in per.get(k, ())) for k, s in shapes.items()}
        scheme = [(s[0], s[1]) for s in case.scheme]   # (sparse fixtures: the plain einsum of every step; the chunked / gathered
        try:                                           #  steps of a sparse scheme never chain, so only their shapes matter here)
            sched = C.triple_schedule(scheme, shapes)
        except Exception as e:   # noqa: BLE001
            print(f"{name} | error: {e}")
            continue
        sh, total, big = dict(shapes), 0.0, 0
        for e in sched:
            i = scheme[e[1]][0][0]
            n_in = numel(sh[i])
            for n in e[1:]:
                (ii, j), eq = scheme[n]
                la, lb, lo = C._labels(eq)
                ext = dict(zip(la, sh[ii]))
                ext.update(zip(lb, sh[j]))
                sh[ii] = tuple(ext[x] for x in lo)
            if n_in >= 2 ** 22:
                total += 8.0 * (n_in + numel(sh[i]))
                big += 1
        tr = [e[1:] for e in sched if e[0] == "triple"]
        print(f"{name} | {big} | {total / 1e9:.1f} | {len(tr)} {tr if tr else ''}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        report()
        sys.exit(0)
    for label, env in (("pairs only (ARTN_FUSE3=0)", {"ARTN_FUSE3": "0"}), ("as shipped", {}),
                       ("64-byte input runs, 96 fragment registers (ARTN_FUSE3_RUN=3 ARTN_FUSE3_FRAG=96)",
                        {"ARTN_FUSE3_RUN": "3", "ARTN_FUSE3_FRAG": "96"})):
        print(f"\n## {label}\n\nfixture | big launches | GB moved by them | triples\n---|---|---|---")
        sys.stdout.flush()
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=dict(os.environ, **env), check=False)
