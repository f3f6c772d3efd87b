"""`from lib.pointops.functions import pointops` (models/PDGNet_v2.py:17) -> pdgn_amd.pointops.

Same module-level names as lib/pointops/functions/pointops.py:30-777 (Function.apply aliases and the grouping
Modules); the CUDA extension `pointops_cuda` is not imported."""
from pdgn_amd.pointops import *  # noqa: F401,F403
from pdgn_amd.pointops import (  # noqa: F401
    Gathering, Gen_QueryAndGroupXYZ, GroupAll, Grouping, Interpolation, KNNQuery, Le_QueryAndGroup,
    Le_QueryAndGroup_OnlyFeature, Le_QueryAndGroup_SameSize, NearestNeighbor, QueryAndGroup, QueryAndGroup_Dilate,
    ballquery, featuredistribute, featuregather, furthestsampling, gathering, grouping, grouping_int, interpolation,
    knnquery, knnquery_exclude, knnquery_naive, labelstat_and_ballquery, labelstat_ballrange, labelstat_idx,
    nearestneighbor, pairwise_distances)
