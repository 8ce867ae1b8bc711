"""flowreg3d_amd -- MI355X (gfx950) engine behind flowreg3d's get_displacement / imregister /
executor API.  See DESIGN.md and INTEGRATION.md."""
from .core import (add_boundary, expand_weight, get_displacement, get_displacement_verify,  # noqa: F401
                   get_motion_tensor_gc,
                   imregister_wrapper, imresize_fused_gauss_cubic3D, level_solver, median_filter5,
                   pyramid_schedule, tensor_factors, warpingDepth)

__version__ = "0.1.0"
