"""Training / evaluation drivers (reference engine/__init__.py)."""
from .defaults import default_argument_parser, default_setup   # noqa: F401
from .launch import launch   # noqa: F401
