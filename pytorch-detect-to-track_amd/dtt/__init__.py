"""MI355X-native Detect-to-Track hot path (host side above the C ABI of libdtt_hip.so).

    from dtt.ops import Correlation, _PSRoIPooling, RoIAlignAvg, _RoIPooling, _RoICrop, nms
    from dtt.rpn import _ProposalLayer, _AnchorTargetLayer

The ops are GPU-only; importing this package does not load the shared library (that happens on first
use, and fails loudly if it has not been built).
"""
__version__ = "0.1.0"
