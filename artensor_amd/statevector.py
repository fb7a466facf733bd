"""Full state-vector evolution on the GPU: the reference's `TensorNetworkCircuit.state_vec`
(artensor/circuit.py:155-175) on top of the pairwise contraction engine.

Every gate application is one step `state <- einsum(state, gate)` with the state as operand 0,
i.e. a dense scheme in which all steps share their first operand; `tensor_contraction` then
fuses consecutive gates pairwise into one pass over the 2^n-element state.  Independent of any
contraction order, this is the on-device cross-check of a tensor-network result for circuits
whose state fits HBM (n <= 33 in complex64 on 288 GB).
"""
import torch

from . import _native as N
from .contraction import tensor_contraction

__all__ = ["state_vec"]


def state_vec(gates, n_qubits, dtype=torch.complex64, device="cuda"):
    """gates: the circuit's gate tensors in order, each `(array, inds)` or an object with
    `.array` / `.inds` (the reference's `circuits_tn[n:]`); `inds` are bond labels "layer-qubit",
    input bonds "0-q".  Returns the state as a tensor of shape (2,)*n with dim q = qubit q
    (a permuted view, like the reference's return value)."""
    pairs = [(g.array, list(g.inds)) if hasattr(g, "inds") else (g[0], list(g[1])) for g in gates]
    state = torch.zeros(2 ** n_qubits, dtype=dtype, device=device)
    state[0] = 1
    tensors = {0: state.reshape((2,) * n_qubits)}
    N.require_gpu(tensors[0], "state_vec")
    inds = [f"0-{q}" for q in range(n_qubits)]
    scheme = []
    for g, (array, ginds) in enumerate(pairs):
        tensors[g + 1] = torch.as_tensor(array).to(dtype).to(device).contiguous()
        gone = [x for x in inds if x in ginds]
        new = [x for x in inds + ginds if x not in gone]     # circuit.py:162-166
        scheme.append(((0, g + 1), (tuple(inds), tuple(ginds), tuple(new))))
        inds = new
    out = tensor_contraction(tensors, scheme) if scheme else tensors[0]
    order = sorted(range(n_qubits), key=lambda d: int(str(inds[d]).split("-")[1]))  # circuit.py:173-175
    return out.permute(tuple(order))
