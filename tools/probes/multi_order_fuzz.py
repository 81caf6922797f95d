"""Randomised comparison of the level-by-level multi-order builder (pp_multiorder_*) with the generic kernels (pp_temporal_* / pp_linegraph_* /
pp_coalesce_*) on streams of random shape: every layer tensor must be equal, bit for bit (tests/test_gpu_multiorder.py: fuzz_against_generic).
usage: multi_order_fuzz.py [cases [seed [max_events]]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathpyg_amd as pp
from tests.test_gpu_multiorder import fuzz_against_generic

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_events = int(float(sys.argv[3])) if len(sys.argv) > 3 else 60_000
t0 = time.time()
taken, back = fuzz_against_generic(pp, cases, seed, max_events, verbose=True)
print(f"{cases} cases equal; level-by-level builder took {taken}, handed {back} back to the generic kernels; {time.time() - t0:.0f} s")
