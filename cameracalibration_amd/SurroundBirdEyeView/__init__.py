"""Drop-in for the reference's SurroundBirdEyeView package (SurroundBirdEyeView/__init__.py:1)."""
from .surroundBEV import BevGenerator  # noqa: F401
