"""`from evaluation.StructuralLosses.nn_distance import nn_distance` (evaluation/evaluation_metrics.py:10)."""
from pdgn_amd.structural_losses.nn_distance import (NNDistance, NNDistanceFunction,  # noqa: F401
                                                    NNDistanceGrad, nn_distance)
