#!/usr/bin/env python3
"""Replays ONE case of the randomised parity tool (the line a soak prints for it: rows cols scene seed {settings}) against the oracle, with
the product's library or an experimental build (BPVO_AB_LIB, scripts/build_exp.sh) — which build a pose difference came in with.

  python tests/tools/replay_case.py 61 213 0 898244736 "{'descriptor': 'intensity', ...}"
"""
import ast
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import bpvo_amd
    from bpvo_amd import capi
    import __graft_entry__ as ge
    import fuzz_parity as fz
    rows, cols, scene, seed = (int(v) for v in sys.argv[1:5])
    kw = ast.literal_eval(sys.argv[5])
    lib = os.environ.get("BPVO_AB_LIB")
    hip = capi.Binding(os.path.join(ROOT, lib), "bpvo_hip_") if lib else bpvo_amd.load()
    orc = capi.Binding(ge.ORACLE_LIB, "bpvo_orc_")
    try:
        print(lib or "product", "->", fz.check(hip, orc, rows, cols, kw, scene, seed), flush=True)
    except AssertionError as e:
        print(lib or "product", "-> FAIL", e.args, flush=True)


if __name__ == "__main__":
    main()
