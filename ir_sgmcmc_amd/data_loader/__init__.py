from .synthetic import init_var_params, synthetic_pair  # noqa: F401
from . import data_loaders  # noqa: F401
from .data_loaders import BiobankDataLoader, SyntheticDataLoader  # noqa: F401
