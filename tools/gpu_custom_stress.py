"""Open-ended randomized parity stress of the custom-pattern path (tests/stressgen.py's custom_batch: host splitter +
external chunk boundaries in the tile kernel) against the Python oracle running the same pattern on PCRE2; the driver-run suite
holds a fixed block of its seeds (tests/test_gpu_stress.py).   python tools/gpu_custom_stress.py --seconds 120 --seed 1   (the lines in profiles/*_stress.txt are this command's last line)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
from test_gpu_stress import check_custom
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0, help="wall-clock budget")
ap.add_argument("--seed", type=int, default=1, help="first seed")
args = ap.parse_args()
budget, seed = args.seconds, args.seed
seed_first = seed
t0 = time.time(); runs = 0; bad = 0
while time.time() - t0 < budget:
    err = check_custom(seed)
    if err:
        bad += 1; print("MISMATCH seed", seed, err, flush=True)
    runs += 1; seed += 1
print(f"{runs} custom-pattern batches, {bad} mismatches (seeds {seed_first}..{seed - 1})")
