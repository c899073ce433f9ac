"""kdtree.New timing probe: python tools/build_probe.py [n]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcgol_amd import synth, kdtree, _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L.check(L.lib().pcgx_init(0))
b = synth.uniform_cloud(n, 10.0, 2)
kdtree.New(b)
t0 = time.perf_counter()
for _ in range(5):
    t = kdtree.New(b); del t
print("build+free n=%d: %.2f ms" % (n, (time.perf_counter() - t0) / 5 * 1e3))
