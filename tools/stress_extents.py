#!/usr/bin/env python3
"""Randomised stress of the EXTENT paths on the GPU (artn_k_xgemm with and without its tail launch, artn_k_xrow, the strided
fallback, split-K of few-tile steps): single steps whose labels have extents 2..9 in random orders, optional batch label,
against torch.einsum in complex128 on the device.   python tools/stress_extents.py [cases] [seed]

Three families of shapes so that every path is hit: general (a few labels per side), row-streaming (tens of thousands of rows,
at most 48 contracted values and 48 columns, the result's fastest label a row label), many-tile (thousands of 128-row tiles
and 4+ column blocks: the two-launch plan)."""
import collections, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import artensor_amd as A

EXT = [2, 3, 4, 5, 6, 7, 9]


def draw(rng, prefix, lo, hi, n_min=1, n_max=6, ext=EXT):
    """Labels whose extents multiply to lo..hi."""
    for _ in range(1000):
        d = {f"{prefix}{i}": int(rng.choice(ext)) for i in range(int(rng.integers(n_min, n_max + 1)))}
        p = int(np.prod(list(d.values())))
        if lo <= p <= hi:
            return d
    raise RuntimeError("no draw")


def rnd(gen, shape):
    x = torch.randn(tuple(shape) + (2,), device="cuda", generator=gen, dtype=torch.float32)
    return torch.view_as_complex(x).contiguous()


def main(cases=300, seed=0):
    rng = np.random.default_rng(seed)
    gen = torch.Generator(device="cuda").manual_seed(seed)
    seen, worst, t0 = collections.Counter(), 0.0, time.time()
    for case in range(cases):
        fam = ("general", "rows", "tiles")[case % 3]
        H = {}
        if fam == "general":
            M, Nn, K = draw(rng, "m", 8, 1 << 16), draw(rng, "n", 1, 4096, 0, 4), draw(rng, "k", 1, 2048, 0, 4)
            if rng.random() < 0.3:
                H = draw(rng, "h", 2, 9, 1, 1)
        elif fam == "rows":
            M, Nn, K = draw(rng, "m", 1 << 15, 1 << 18, 3, 9), draw(rng, "n", 1, 48, 1, 3), draw(rng, "k", 1, 48, 1, 3)
        else:
            M, Nn, K = draw(rng, "m", 1 << 17, 1 << 19, 3, 9), draw(rng, "n", 100, 400, 2, 4), draw(rng, "k", 2, 64, 1, 3)
        ext = {**M, **Nn, **K, **H}
        la, lb, lo = list(M) + list(K) + list(H), list(Nn) + list(K) + list(H), list(M) + list(Nn) + list(H)
        for lst in (la, lb, lo):
            rng.shuffle(lst)
        if fam == "rows" and rng.random() < 0.8:   # the result's fastest label a row label
            last_m = max(i for i, x in enumerate(lo) if x in M)
            lo[last_m], lo[-1] = lo[-1], lo[last_m]
        sa, sb = tuple(ext[x] for x in la), tuple(ext[x] for x in lb)
        if np.prod([float(v) for v in ext.values()]) > 6e9:
            continue
        eq = (tuple(la), tuple(lb), tuple(lo))
        info = A.step_info(eq, sa, sb)
        rows_, cols_ = int(np.prod(list(M.values()))), int(np.prod(list(Nn.values())))
        if rows_ < cols_:
            rows_, cols_ = cols_, rows_
        hp = int(np.prod(list(H.values()))) if H else 1
        if info["kernel"] == 5 and info["m_tile_bits"] in (4, 6):
            path = "xrow" if info["m_tile_bits"] == 4 else "xrow64"
        elif info["kernel"] == 5:   # (the planner's rule for the second launch, restated: 4+ column blocks, not a multiple of 3, 8+ main tiles per CU)
            blocks, rt = -(-cols_ // 32), -(-rows_ // 128) * hp
            path = "xgemm+tail" if blocks >= 4 and blocks % 3 and rt * (blocks // 3) >= 2048 else "xgemm"
        else:
            path = {0: "strided"}.get(info["kernel"], f"kernel{info['kernel']}")
        a, b = rnd(gen, sa), rnd(gen, sb)
        got = A.contract(eq, a, b)
        sym = {x: chr(65 + i) if i < 26 else chr(97 + i - 26) for i, x in enumerate(ext)}
        want = torch.einsum("".join(sym[x] for x in la) + "," + "".join(sym[x] for x in lb) + "->" + "".join(sym[x] for x in lo),
                            a.to(torch.complex128), b.to(torch.complex128))
        err = float((got.to(torch.complex128) - want).abs().max() / want.abs().max())
        tol = 3e-6 * max(1.0, float(np.prod(list(K.values()))) / 256.0) ** 0.5
        if not err <= tol:
            print("FAIL", case, fam, eq, sa, sb, err, tol, info)
            return 1
        worst = max(worst, err)
        seen[(fam, path)] += 1
    print(f"{cases} cases, seed {seed}: worst error {worst:.2e} of the largest result; paths {dict(seen)}; {time.time() - t0:.0f} s")   # (einsum and contract run on the device)
    return 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
