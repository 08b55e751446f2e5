"""rfdnet_amd -- MI355X (gfx950) native hot path of RfD-Net.

PointNet++ set-abstraction operators (FPS, ball query, grouping, three_nn
interpolation) and the fused occupancy decoder as hand-written HIP kernels
behind the reference's own operator / module interfaces.  See DESIGN.md.
"""
__version__ = "0.1.0"

import os as _os

# ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in the order of their first use, and streams that
# share a queue serialise: with several scenes in flight per GPU (one compute + one copy stream each) a 30-ms mesh blit then
# holds up another scene's kernels (2 queues -8 %, 16 queues +3-5 % at four scenes in flight: profiles/r04_hw_queues.txt).
# The runtime reads the variable when it initialises, so the package sets a default at import -- a host application no
# longer has to know; an explicit setting of the caller wins, RFD_KEEP_HW_QUEUES=1 opts out,
# and if HIP is already up this has no effect.
if not _os.environ.get("RFD_KEEP_HW_QUEUES"):
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
