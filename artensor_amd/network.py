"""`NumericalTensorNetwork.contract` (reference artensor/tensor_network.py:207-226) on the device.

The reference simplifies a freshly built network (`_simplify`, tensor_network.py:92-135) by
contracting dangling vectors, matrices and multi-bond neighbours pairwise with `torch.einsum`.
`tn_contract(tn, x, y)` is that pairwise step with the einsum replaced by the contraction
engine; the network object is consumed by duck typing (`tensor_bonds`, `bond_tensors`,
`tensors`), so a maintainer can bind it in place of the method:

    NumericalTensorNetwork.contract = artensor_amd.tn_contract
"""
from .contraction import contract

__all__ = ["tn_contract"]


def tn_contract(tn, x, y):
    """Contract tensors x and y of `tn` into x (tensor_network.py:207-226): shared bonds are
    summed, the others kept in the order (bonds of x, then bonds of y)."""
    if x not in tn.tensor_bonds or y not in tn.tensor_bonds:
        raise RuntimeError(f"tensor ids {x}, {y} must both be in the network")
    bonds_x, bonds_y = tn.tensor_bonds.pop(x), tn.tensor_bonds.pop(y)
    contracted = [b for b in bonds_x if b in bonds_y]
    bonds_new = [b for b in bonds_x + bonds_y if b not in contracted]
    for b in contracted:
        tn.bond_tensors.pop(b)
    for b in [b for b in bonds_y if b not in contracted]:
        tn.bond_tensors[b].remove(y)
        tn.bond_tensors[b].add(x)
    tn.tensor_bonds[x] = bonds_new
    tn.tensors[x] = contract((tuple(bonds_x), tuple(bonds_y), tuple(bonds_new)), tn.tensors.pop(x), tn.tensors.pop(y))
    return tn.tensors[x]
