"""allset_amd -- MI355X-native implementation of AllSet's vertex<->hyperedge multiset aggregation
(HalfNLHconv / PMA) behind the reference's own module surface.  HIP kernels live in ``csrc/`` behind
the C ABI of ``include/allset_hip.h``; there is no CPU or eager fallback on the aggregation path."""
from .incidence import Incidence, cached_incidence          # noqa: F401
from .functional import deepsets_aggregate, pma_aggregate, pma_attention_weights   # noqa: F401
from .layers import MLP, PMA, HalfNLHconv, glorot, zeros    # noqa: F401
from .models import SetGNN                                  # noqa: F401

__version__ = "0.1.0"
