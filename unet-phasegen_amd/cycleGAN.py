"""``demo.py:5`` and ``data.py:67`` of the reference import UNetModel from a module named ``cycleGAN`` that is not in
the reference repository; this alias makes those imports resolve."""
from phasegen.model import UNetModel  # noqa: F401
