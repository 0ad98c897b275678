"""Open-ended randomized parity stress (tests/stressgen.py's random_batch; the driver-run suite holds a
fixed block of its seeds in tests/test_gpu_stress.py).   python tools/gpu_stress.py --seconds 120 --seed 1000   (the lines in profiles/*_stress.txt are this command's last line)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as tg
from stressgen import random_batch, MODES_ALL
from test_gpu_stress import _compiled
from oracle.coracle import COracle
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0, help="wall-clock budget")
ap.add_argument("--seed", type=int, default=1000, help="first seed")
ap.add_argument("--device", action="store_true", help="every batch ALSO device-resident through spl_encode_batch_device (batches of up to 1536 tiles: the one-launch form), "
                "encoded twice -- the chunk memo the handle has built over the earlier batches, then with this batch's chunks in it")
args = ap.parse_args()
budget, seed0 = args.seconds, args.seed
orcs = {}
def coracle(name):
    if name not in orcs: orcs[name] = COracle(name)
    return orcs[name]
MODES = [m for m in MODES_ALL if m in _compiled(sorted(set(MODES_ALL)))]
t0 = time.time(); runs = 0; bad = 0; seed = seed0
while time.time() - t0 < budget:
    name, geom, special, texts = random_batch(seed, MODES)
    tg._force_tiles(name, geom)
    try:
        tg.assert_batch_equal(name, texts, coracle, special=special)
        if args.device and texts:
            import numpy as np, torch
            from splintr_amd.device import DeviceBatch, encode_device, result_csr
            o_ids, o_off = tg.oracle_csr(coracle(name), texts, special)
            b = DeviceBatch(texts, torch.device("cuda", 0))
            for rep in range(2):
                encode_device(tg.tok(name), b, with_special=special); torch.cuda.synchronize()
                ids, off = result_csr(b)
                assert np.array_equal(off, o_off) and np.array_equal(ids, o_ids), f"device-resident pass {rep}"
    except AssertionError as e:
        bad += 1; print("MISMATCH seed", seed, name, "geom", geom, "special", special, str(e)[:300], flush=True)
    finally:
        tg._force_tiles(name, 0)
    runs += 1; seed += 1
print(f"{runs} random batches, {bad} mismatches (seeds {seed0}..{seed - 1})")
