"""Dev aid: runs of one class that outgrow a tile's halo and end in multi-byte characters of the same class, at every
alignment against the window edges -- the family of tools/dev/gpu_stress.py's seed 22739 (every vocabulary and mode)."""
import json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as tg
from oracle.coracle import COracle
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
orcs = {}
def coracle(name):
    if name not in orcs: orcs[name] = COracle(name)
    return orcs[name]
FAMILIES = [("0", ["Ⅷ", "٣", "½", "\U0001d7d8"]), ("a", ["é", "你", "ǅ", "\U00010400"]),
            ("A", ["É", "Ж", "你"]), (" ", [" ", " ", "　"]),
            ("-", ["—", "。", "\U0001f642", "§"]), ("\n", [" ", ""]), ("1a", ["Ⅷé"]),
            ("x'", ["'ſ", "’s"]), ("你", ["好", "。"])]
lits = json.load(open(os.path.join(ROOT, "splintr_amd", "data", "special_tokens.json"), encoding="utf-8"))
pad = lambda n: ("lorem ipsum " * 400)[:n]
t0 = time.time(); runs = 0; bad = 0; seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
while time.time() - t0 < budget:
    rng = random.Random(seed)
    name = rng.choice(tg.VOCABS); geom = rng.choice([0, 0, 1, 3, 4, 5, 5])
    unit, tails = rng.choice(FAMILIES)
    L = rng.choice([90, 120, 130, 160, 190, 200, 230, 240, 260, 400, 700, 1100])
    head = rng.choice(["", rng.choice(tails), rng.choice(tails) * 3])
    tail = "".join(rng.choice(tails) for _ in range(rng.randint(1, 3)))
    after = rng.choice([" and the end", "\nx", "", "!", "9", "Z"])
    special = rng.random() < 0.3                              # a special token right behind (or inside) the run's end
    if special:
        lit = rng.choice(list(lits[name]))
        after = rng.choice([lit, lit + after, after + lit]); tail = rng.choice([tail, tail + lit + tail])
    k0 = rng.randrange(0, 900)
    texts = [pad(k) + head + unit * (L // len(unit)) + tail + after for k in range(k0, k0 + 300)]
    mix = rng.random()
    if mix < 0.3: texts = ["".join(texts)]
    elif mix < 0.5: texts = [x for t in texts for x in (t, rng.choice(["", "", "a", "é"]))]   # empty / tiny texts behind the edge
    tg._force_tiles(name, geom)
    try:
        tg.assert_batch_equal(name, texts, coracle, special=special)
    except AssertionError as e:
        bad += 1; print("MISMATCH seed", seed, name, "geom", geom, "special", special, ascii(unit), L, ascii(head), ascii(tail), ascii(after), ascii(str(e)[:200]), flush=True)
    finally:
        tg._force_tiles(name, 0)
    runs += 1; seed += 1
print(f"{runs} edge batches, {bad} mismatches")
