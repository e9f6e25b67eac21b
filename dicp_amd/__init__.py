"""dicp_amd -- MI355X-native differentiable ICP behind utiasASRL/dICP's call surface.

    from dicp_amd.ICP import ICP          # == dICP.ICP.ICP
    from dicp_amd.nn import nn            # == dICP.nn.nn
    from dicp_amd.loss import loss        # == dICP.loss.loss

The per-iteration hot path runs in libdicp_hip.so (hand-written gfx950 kernels behind
include/dicp_hip.h).  There is no CPU compute path.
"""
__version__ = "0.1.0"
