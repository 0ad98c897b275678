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
    """gen_c5_pieces: the ranks' pieces of BASELINE config 5 concatenate to the 100 documents, and every cut
    inside a document sits behind a newline, in front of an ASCII letter or digit (a context-free match
    boundary of every split pattern)."""
    sys.path.insert(0, ROOT)
    import bench
    saved = (bench.C5_DOCS, bench._c5_doc)
    try:
        from splintr_amd import corpus
        bench.C5_DOCS = 5
        docs = {k: corpus.c5(1, seed=1005 + k, doc_bytes=1 << 15)[0] for k in range(5)}
        for world in (1, 2, 4, 8):
            pieces = []

            class _P:                       # (no fork in the test: map in-process)
                def __init__(self, *a): pass
                def __enter__(self): return self
                def __exit__(self, *a): return False
                def map(self, f, it): return [docs[k] for k in it]
            import multiprocessing
            real = multiprocessing.Pool
            multiprocessing.Pool = _P
            try:
                for r in range(world):
                    pieces.append(bench.gen_c5_pieces(r, world))
            finally:
                multiprocessing.Pool = real
            flat = "".join(p for ps in pieces for p in ps)
            assert flat == "".join(docs[k] for k in range(5)), world
            # cuts inside documents: the piece starts with an ASCII alnum and the previous piece ends with \n
            total, starts = 0, set()
            for k in range(5):
                starts.add(total)
                total += len(docs[k])
            pos = 0
            prev = None
            for ps in pieces:
                for p in ps:
                    if pos not in starts:
                        assert prev is not None and prev.endswith("\n") and p[0].isascii() and p[0].isalnum(), (world, pos)
                    pos += len(p)
                    prev = p
    finally:
        bench.C5_DOCS, bench._c5_doc = saved
