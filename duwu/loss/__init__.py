"""reference src/duwu/loss/__init__.py."""
from uwudiff_amd.objective import (DiffusionLoss, DiffusionLossAuxOutput, NNWeightedRFLoss,  # noqa: F401
                                   NNWeightedRFLossAuxOutput, RectifiedFlowLoss)
