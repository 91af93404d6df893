#!/usr/bin/env python3
"""Replays fuzz cases given as 'rows cols scene seed {kw}' lines (the text after FAIL / EXCEPTION in a fuzz log)."""
import ast
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import bpvo_amd  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
from bpvo_amd import capi  # noqa: E402
import fuzz_parity as fz  # noqa: E402

hip = bpvo_amd.load()
orc = capi.Binding(ge.ORACLE_LIB, "bpvo_orc_")
for line in open(sys.argv[1]):
    line = line.strip()
    if not line or line.startswith("#"):
        continue
    head, brace = line.split("{", 1)
    rows, cols, scene, seed = (int(v) for v in head.split()[-4:])
    kw = ast.literal_eval("{" + brace.split("}", 1)[0] + "}")
    try:
        print(rows, cols, scene, seed, kw["descriptor"], "->", fz.check(hip, orc, rows, cols, kw, scene, seed), flush=True)
    except AssertionError as e:
        print(rows, cols, scene, seed, kw["descriptor"], "-> FAIL", e.args, flush=True)
