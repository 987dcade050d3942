"""CPU oracle (test infrastructure only -- see oracle/ops.py header).  Never imported by the product."""
from . import ops  # noqa: F401
from .transition import OracleChain, OracleConfig  # noqa: F401
