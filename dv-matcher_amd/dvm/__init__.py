"""dvm — host-side binding of libdvm_hip.so (the MI355X kernels of DV-Matcher's
correspondence hot path).  The reference-named modules `models.model`, `models.loss`
and `lib.deformation_graph_point` in this directory are thin layers over `dvm.ops`.
"""
from . import _lib  # noqa: F401
