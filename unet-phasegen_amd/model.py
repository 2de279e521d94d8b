"""Drop-in for the reference's ``model.py``: ``from model import UNetModel`` (train.py:6)."""
from phasegen.model import UNetModel  # noqa: F401
