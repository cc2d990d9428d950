from .build import build_model  # noqa: F401
