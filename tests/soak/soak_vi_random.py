"""soak of tests/test_gpu_vi.py::test_random_hybrid_graphs_through_every_factor_kernel: many more seeds than the suite runs (random hybrid
graphs through the tiny-grid, the group and the thread-per-factor kernels against the C oracle and against each other).
usage: python tests/soak/soak_vi_random.py [first seed] [count]"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd'), os.path.join(ROOT, 'tests')]
spec = importlib.util.spec_from_file_location('t', os.path.join(ROOT, 'tests', 'test_gpu_vi.py'))
t = importlib.util.module_from_spec(spec)
spec.loader.exec_module(t)
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 1000), (int(sys.argv[2]) if len(sys.argv) > 2 else 60)
fn = t._random_hybrid_check
ok, skipped, t0 = 0, 0, time.time()
for i in range(count):
    seed = first + i
    K, T, quirks = 1 + seed % 2, (2, 3, 5, 4)[(seed // 2) % 4], bool((seed // 8) % 2)
    try:
        if fn(seed, K, T, quirks, require_every_kernel=False) == 'overflow':
            skipped += 1              # (a formula overflowed on the grid: the reference itself raises there)
        ok += 1
    except AssertionError as e:
        print('FAIL seed %d K %d T %d quirks %s: %s' % (seed, K, T, quirks, str(e)[:400]), flush=True)
print('%d of %d seeds pass, %d of them skipped as invalid inputs (%.0f s)' % (ok, count, skipped, time.time() - t0))
sys.exit(0 if ok == count else 1)
