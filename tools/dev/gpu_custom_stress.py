"""Dev aid: open-ended randomized parity stress of the custom-pattern path (tests/stressgen.py's custom_batch: host splitter +
external chunk boundaries in the tile kernel) against the Python oracle running the same pattern on PCRE2; the driver-run suite
holds a fixed block of its seeds (tests/test_gpu_stress.py).   python tools/dev/gpu_custom_stress.py [seconds] [first seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
from test_gpu_stress import check_custom
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time(); runs = 0; bad = 0
while time.time() - t0 < budget:
    err = check_custom(seed)
    if err:
        bad += 1; print("MISMATCH seed", seed, err, flush=True)
    runs += 1; seed += 1
print(f"{runs} custom-pattern batches, {bad} mismatches")
