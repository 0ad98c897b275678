"""`python bench.py --gpus N` as a plain command (no WORLD_SIZE in the environment) must launch the ranks
itself -- the driver's scaling run invokes it that way.  The launcher logic is exercised here on CPU with
the hidden --launch-selftest flag: bench.py re-executes itself under torch.distributed.run, the ranks
rendezvous on gloo at 127.0.0.1, rank 0 prints the one JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_self_launch_two_ranks_gloo():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-selftest"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out == {"launcher": "ok", "n_gpus": 2, "rank_sum": 3}


def test_more_gpus_than_the_box_has_is_a_clear_error():
    """On a box with fewer GPUs than asked for the command must say so -- not die in the launcher."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 2), "--steps", "2", "--warmup", "1"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert f"needs {have + 2} GPUs, this box has {have}" in (p.stderr + p.stdout)


def test_c5_pieces_tile_the_documents():
    """bench.py's BASELINE config 5 in a distributed run: every document is cut into pieces whose concatenation is the document, every
    cut sits behind a newline in front of an ASCII letter or digit (a context-free match boundary of every built-in split pattern),
    and the ranks' slices of the waves, read wave by wave and rank by rank, are the pieces in document order."""
    sys.path.insert(0, ROOT)
    import bench
    from splintr_amd import corpus
    saved = (bench.C5_DOCS, bench._c5_doc)
    docs = {k: corpus.c5(1, seed=1005 + k, doc_bytes=1 << 15)[0] for k in range(5)}
    try:
        bench.C5_DOCS = 5
        bench._c5_doc = lambda k: docs[k]
        allp = []
        for k in range(5):
            ps = bench._c5_doc_pieces(k)
            assert len(ps) == bench.C5_PIECES and "".join(ps) == docs[k]
            for a, b in zip(ps, ps[1:]):
                if b:
                    assert a.endswith("\n") and b[0].isascii() and b[0].isalnum(), (k, a[-5:], b[:5])
            allp += ps

        class _P:                       # (no fork in the test: map in-process)
            def __init__(self, *a): pass
            def __enter__(self): return self
            def __exit__(self, *a): return False
            def map(self, f, it): return [f(x) for x in it]
        import multiprocessing
        real = multiprocessing.Pool
        multiprocessing.Pool = _P
        os.environ["SPL_BENCH_FORCE_DIST"] = "1"
        try:
            for world in (1, 2, 3, 8):
                waves = [bench.gen_c5_pieces(r, world) for r in range(world)]          # [rank][wave] -> pieces
                n_waves = len(waves[0])
                flat = [p for k in range(n_waves) for r in range(world) for p in waves[r][k]]
                assert flat == allp, world
        finally:
            multiprocessing.Pool = real
            os.environ.pop("SPL_BENCH_FORCE_DIST", None)
        sl = bench.c5_wave_slices(100, 8, bench.N_WAVES)
        assert [x for k in range(bench.N_WAVES) for r in range(8) for x in range(*sl[k][r])] == list(range(100 * bench.C5_PIECES))
        # tapered: the waves shrink, the last one is a few per cent of the batch; within a wave the ranks' slices differ by at most one piece
        sizes = [sum(b - a for a, b in sl[k]) for k in range(bench.N_WAVES)]
        assert sizes == sorted(sizes, reverse=True) and sizes[-1] <= 0.08 * sum(sizes) and sizes[0] >= 0.18 * sum(sizes), sizes
        assert all(max(b - a for a, b in sl[k]) - min(b - a for a, b in sl[k]) <= 1 for k in range(bench.N_WAVES))
        eq = bench.c5_wave_slices(100, 8, 8, taper=1.0)
        assert max(b - a for k in range(8) for a, b in eq[k]) == min(b - a for k in range(8) for a, b in eq[k]) == 25
        # config 4's prompts: every rank's slices of every wave tile the million prompts in order
        cs = [bench.c4_wave_slices(r, 8) for r in range(8)]
        assert [x for k in range(bench.N_WAVES) for r in range(8) for x in (cs[r][k][0], cs[r][k][1])][0] == 0
        ends = [cs[r][k] for k in range(bench.N_WAVES) for r in range(8)]
        assert all(a[1] == b[0] for a, b in zip(ends, ends[1:])) and ends[-1][1] == bench.C4_PARTS * bench.C4_PART_DOCS
    finally:
        bench.C5_DOCS, bench._c5_doc = saved
