#!/usr/bin/env python
"""Run a script of this repo against a measurement build of the library:  python tools/probes/with_so.py <libgpn_x.so> <script.py> [args]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gapartnet_amd import _C

_C.SO_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
