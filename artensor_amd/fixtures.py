"""Single-file (.npz) container for a contraction case.

A *case* is what travels from the planning side (the reference's Python planner,
``/root/reference/artensor/order_finder.py:174`` + ``contraction.py:23/208``) to the
execution side: leaf tensors, the compiled scheme (dense 2-tuples or sparse 3/5-tuples,
exactly the step formats of ``contraction.py:52`` and ``contraction.py:327-335``),
optional slicing indices (``simulation.py:60-65``) and expected outputs.

The format is plain numpy so it can be produced on a box that has the reference and
consumed on a GPU box that has not.
"""
import json
import numpy as np
import torch


def save_case(path, tensors, scheme, meta=None, arrays=None, slicing_indices=None):
    """tensors: dict/list id -> torch/numpy array; scheme: reference-format step list."""
    out = {}
    if isinstance(tensors, (list, tuple)):
        tensors = {i: t for i, t in enumerate(tensors)}
    ids = sorted(tensors.keys())
    for i in ids:
        t = tensors[i]
        if isinstance(t, torch.Tensor):
            t = t.detach().cpu().numpy()
        out[f"t{i}"] = np.ascontiguousarray(t)
    steps = []
    for n, step in enumerate(scheme):
        rec = {"edge": [int(step[0][0]), int(step[0][1])], "eq": step[1], "len": len(step)}
        if len(step) >= 3:
            bi, bj = step[2]
            rec["nbi"], rec["nbj"] = len(bi), len(bj)
            for k, idx in enumerate(bi):
                out[f"s{n}_bi{k}"] = np.asarray(idx, dtype=np.int64)
            for k, idx in enumerate(bj):
                out[f"s{n}_bj{k}"] = np.asarray(idx, dtype=np.int64)
        if len(step) == 5:
            rec["rshape"] = None if step[3] is None else [int(x) for x in step[3]]
            rec["next_shape"] = [int(x) for x in step[4]]
        steps.append(rec)
    m = dict(meta or {})
    m["tensor_ids"] = [int(i) for i in ids]
    m["steps"] = steps
    if slicing_indices is not None:
        # ordered list of [bond_label, [[tid, dim], ...]] -- order matters (simulation.py:108)
        m["slicing_indices"] = [
            [str(b), [[int(t), int(d)] for t, d in v]] for b, v in slicing_indices.items()
        ]
    out["meta"] = np.frombuffer(json.dumps(m).encode(), dtype=np.uint8)
    for k, v in (arrays or {}).items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[f"x_{k}"] = np.asarray(v)
    np.savez_compressed(path, **out)


class Case:
    def __init__(self, path):
        z = np.load(path, allow_pickle=False)
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.tensors = {i: torch.from_numpy(z[f"t{i}"]) for i in self.meta["tensor_ids"]}
        self.arrays = {k[2:]: z[k] for k in z.files if k.startswith("x_")}
        self.scheme = []
        for n, rec in enumerate(self.meta["steps"]):
            edge = (rec["edge"][0], rec["edge"][1])
            if rec["len"] == 2:
                self.scheme.append((edge, rec["eq"]))
                continue
            bi = [torch.from_numpy(z[f"s{n}_bi{k}"]) for k in range(rec["nbi"])]
            bj = [torch.from_numpy(z[f"s{n}_bj{k}"]) for k in range(rec["nbj"])]
            if rec["len"] == 3:
                self.scheme.append((edge, rec["eq"], [bi, bj]))
            else:
                rshape = None if rec["rshape"] is None else tuple(rec["rshape"])
                self.scheme.append((edge, rec["eq"], [bi, bj], rshape, tuple(rec["next_shape"])))
        self.slicing_indices = None
        if "slicing_indices" in self.meta:
            self.slicing_indices = {
                b: [(t, d) for t, d in v] for b, v in self.meta["slicing_indices"]
            }

    def fresh_tensors(self, dtype=None, device=None):
        out = {}
        for i, t in self.tensors.items():
            t = t.clone()
            if dtype is not None:
                t = t.to(dtype)
            if device is not None:
                t = t.to(device)
            out[i] = t
        return out


def load_case(path):
    return Case(path)
