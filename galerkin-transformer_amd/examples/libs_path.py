"""Put the package root on sys.path so ``from libs import *`` resolves (same role as the
reference's examples/libs_path.py)."""
import os
import sys

SRC_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if SRC_ROOT not in sys.path:
    sys.path.append(SRC_ROOT)
