"""pgvector's distance hot path on MI355X (gfx950).

The product is the C-ABI shared library `pgvector_amd/lib/libpgv_hip.so`
(include/pgv_hip.h); this package holds its HIP sources (csrc/), the C host
glue that mirrors the reference's index-AM loops (host/), and a ctypes face of
the ABI for the Python test/bench harness.  Importing it never touches
oracle/, and there is no CPU fallback.
"""
from . import api  # noqa: F401
from ._lib import (PGV_F16, PGV_F32, PGV_L1, PGV_L2SQ, PGV_NEG_IP, PGV_OPS_COSINE,  # noqa: F401
                   PGV_OPS_IP, PGV_OPS_L2, PgvError, lib)
