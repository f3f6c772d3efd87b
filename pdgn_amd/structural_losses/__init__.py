"""Drop-in for the reference's ``evaluation/StructuralLosses`` package
(``match_cost``, ``nn_distance``) on libpdgn_hip.so."""
from .match_cost import match_cost, emd_cost  # noqa: F401
from .nn_distance import nn_distance  # noqa: F401
