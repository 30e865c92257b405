"""MI355X-native multi-scalar multiplication behind the ZPrize-2022 prize1-msm operator API.

The directory name (``2022-entries_amd``) is not a Python identifier; import it through the
``entries_amd`` shim at the repo root (``import entries_amd``), which loads this package.
"""
from .msm import (  # noqa: F401
    CURVE_IDS,
    MsmError,
    MultiScalarMultContext,
    VariableBaseMSM,
    affine_stride,
    projective_bytes,
    fold_partials,
    generate_points,
    library_path,
    load_library,
    multi_scalar_mult,
    multi_scalar_mult_init,
    plan,
    msm,
    last_stateless,
    trim,
    pool_stats,
    ChunkedPippenger,
    HashMapPippenger,
)
from .dist import all_gather_partials, shard_bounds, sharded_msm  # noqa: F401,E402
from . import formats  # noqa: F401,E402
