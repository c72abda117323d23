"""one-off soak: tests/test_gpu_edge_cases.py::test_random_shapes_against_oracle for more seeds (usage: fuzz_shapes.py first last)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd'), os.path.join(ROOT, 'tests')]
import test_gpu_edge_cases as t
from lhvi import _abi
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    try:
        t.test_random_shapes_against_oracle(_abi, seed)
    except AssertionError as e:
        bad += 1
        print('seed', seed, 'FAILED', str(e)[:300], flush=True)
print('seeds %d..%d: %d failed' % (lo, hi - 1, bad))
