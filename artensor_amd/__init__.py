"""artensor_amd: MI355X-native numerical contraction engine behind artensor's executor API.

Drop-in for the hot path of Fanerst/artensor:

    from artensor_amd import tensor_contraction, tensor_contraction_sparse   # executors
    from artensor_amd import contraction_scheme, contraction_scheme_sparse   # scheme compilers
    from artensor_amd import TensorNetworkSimulation, sliced_contraction     # slice loop, multi-GPU

    from artensor_amd import tensor_network_contraction, quantum_circuit_simulation   # one-call API

Planning (AbstractTensorNetwork / ContractionTree / find_order / GreedyOrderFinder, the circuit parser)
is not part of this package: the engine consumes the planner's products unchanged; the one-call API and
TensorNetworkSimulation.from_circuit_file / prepare_contraction forward the planning half to the
reference's front end (the installed `artensor` package, or `planner=`).
"""
from .contraction import (  # noqa: F401
    contract,
    contraction_scheme,
    contraction_scheme_sparse,
    einsum_eq_convert,
    precision,
    step_info,
    tensor_contraction,
    tensor_contraction_sparse,
)
from .fixtures import load_case, save_case  # noqa: F401
from .simulation import (  # noqa: F401
    SliceRunner,
    TensorNetworkSimulation,
    accumulate,
    apply_slice,
    partition_output,
    partitioned_contraction,
    plan_output_slabs,
    slab_contraction,
    quantum_circuit_simulation,
    rank_slices,
    slice_assignments,
    sliced_contraction,
    tensor_network_contraction,
)

from .network import tn_contract  # noqa: F401
from .statevector import state_vec  # noqa: F401

__version__ = "0.1.0"
