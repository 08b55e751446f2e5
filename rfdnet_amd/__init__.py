"""rfdnet_amd -- MI355X (gfx950) native hot path of RfD-Net.

PointNet++ set-abstraction operators (FPS, ball query, grouping, three_nn
interpolation) and the fused occupancy decoder as hand-written HIP kernels
behind the reference's own operator / module interfaces.  See DESIGN.md.
"""
__version__ = "0.1.0"
