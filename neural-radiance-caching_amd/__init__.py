"""MI355X-native radiance-cache ray-batch renderer (hot path of
benattal/neural-radiance-caching's Model.__call__ / render_image).

Import as `nrc_amd` (the directory name carries the reference's hyphenated name;
`nrc_amd.py` at the repository root registers this package under that module name).
"""
from .config import GridConfig, RenderConfig, TransientConfig, cornell_transient_config, hotdog_config  # noqa: F401
from .rays import Pixels, Rays, synthetic_rays, synthetic_camera_rays, synthetic_transient_rays  # noqa: F401
from .weights import param_shapes, synthetic_weights  # noqa: F401
from .camera import Camera, cast_ray_batch, cast_spherical_rays, get_pixtocam, render_camera  # noqa: F401,E402
from . import checkpoint, prng, train  # noqa: F401,E402
