"""Dev aid: open-ended sweep of tests/stressgen.py's edge_batch -- runs of one class that outgrow a tile's halo and end in
multi-byte characters of the same class, at every alignment against the window edges (the family of seed 22739).
python tools/dev/gpu_edge_sweep.py [seconds] [first seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as tg
from stressgen import edge_batch, EDGE_MODES
from test_gpu_stress import _compiled
from oracle.coracle import COracle
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
orcs = {}
def coracle(name):
    if name not in orcs: orcs[name] = COracle(name)
    return orcs[name]
MODES = [m for m in EDGE_MODES if m in _compiled(sorted(set(EDGE_MODES)))]
t0 = time.time(); runs = 0; bad = 0; seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
while time.time() - t0 < budget:
    name, geom, special, texts = edge_batch(seed, MODES)
    tg._force_tiles(name, geom)
    try:
        tg.assert_batch_equal(name, texts, coracle, special=special)
    except AssertionError as e:
        bad += 1; print("MISMATCH seed", seed, name, "geom", geom, "special", special, ascii(str(e)[:300]), flush=True)
    finally:
        tg._force_tiles(name, 0)
    runs += 1; seed += 1
print(f"{runs} edge batches, {bad} mismatches")
