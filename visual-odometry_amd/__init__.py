"""visual-odometry_amd: MI355X (gfx950) implementation of the projective-ICP hot
path of lucanunz/Visual-odometry (PICPSolver + appearance matcher +
triangulation), as hand-written HIP kernels behind a C ABI (include/vo_hip.h).

The directory name carries a hyphen, so the package is loaded through
`__graft_entry__.load_package()` under the module name `visual_odometry_amd`.
"""
from . import synth  # noqa: F401  (data generation only)
from .pipeline import BatchPipeline, FramePipeline, SequencePipeline, frames_batch_ragged, match_batch_ragged  # noqa: F401
from .api import (  # noqa: F401
    Camera,
    Context,
    Event,
    KdTree,
    Map,
    PICPSolver,
    VoError,
    compute_correspondences_images,
    default_context,
    estimate_transform,
    extract_correspondences_world,
    load_library,
    radius_search,
    transform_points,
    triangulate_points,
    LIB_PATH,
)
