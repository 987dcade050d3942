from .synthetic import init_var_params, synthetic_pair  # noqa: F401
