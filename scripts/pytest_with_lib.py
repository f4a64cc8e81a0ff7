"""Run pytest in this process against another build of the library: python scripts/pytest_with_lib.py <lib.so> <pytest args>."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import depth_estimation_amd._lib as L

L.LIB_PATH = os.path.abspath(sys.argv[1])
import pytest

sys.exit(pytest.main(sys.argv[2:]))
