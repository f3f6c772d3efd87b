"""`from evaluation.StructuralLosses.match_cost import match_cost` (evaluation/evaluation_metrics.py:9)."""
from pdgn_amd.structural_losses.match_cost import (ApproxMatch, MatchCost, MatchCostFunction,  # noqa: F401
                                                   MatchCostGrad, match_cost)
