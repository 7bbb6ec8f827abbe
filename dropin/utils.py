"""Flat-import shim: the reference's nodes import `utils` from their script directory
(position_finder.py:8-20). Put this directory ahead of the reference's on sys.path / PYTHONPATH."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_matching_amd.utils import *  # noqa: F401,F403,E402
