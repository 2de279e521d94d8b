"""Drop-in for the hot-path part of the reference's ``utils.py``: ``from utils import generate_audio`` (train.py:5,
demo.py:4).  Plotting helpers and the unused cycleGAN leftovers (utils.py:46-83,136-262) are out of scope."""
from phasegen.audio import generate_audio  # noqa: F401
from phasegen.audio import griffin_lim  # noqa: F401,E402
