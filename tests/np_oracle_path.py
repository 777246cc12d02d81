"""Puts oracle/ on sys.path so tests can `import np_oracle` (test infrastructure only)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
