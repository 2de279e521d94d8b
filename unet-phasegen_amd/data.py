"""Drop-in for the reference's ``data.py``: ``from data import get_fft_npy_loader`` (train.py:8, demo.py:3)."""
from phasegen.data import get_fft_npy_loader, get_spec_and_angle  # noqa: F401
