"""pdgn_amd -- MI355X-native implementation of PDGN's data-parallel hot path.

Sub-modules mirror the reference's plugin boundary:
  pointops            lib/pointops/functions/pointops.py  (knnquery, grouping, interpolation, ...)
  structural_losses   evaluation/StructuralLosses/{match_cost,nn_distance}.py
  deconv / generator  models/PDGNet_v2.py point-deconvolution stack (PointDeconv, PointGenerator, D1-4)
  losses / trainer    utils/chamfer_loss.py, PDGNet_v2.get_local_pair and the G+D iteration
All device code is hand-written HIP for gfx950 behind the C ABI in include/pdgn_hip.h.
"""
__version__ = "0.1.0"
