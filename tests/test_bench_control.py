"""bench.py's control plane of last resort (files on the node) with two real processes: barrier + max-over-ranks work
without torch, RCCL or a GPU, and clean up after themselves."""
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, base, q):
    sys.path.insert(0, ROOT)
    sys.argv = ["bench.py"]
    import bench
    from nx_signal_amd import _lib

    f = bench.FileControl(_lib.load(), base, 2, rank)
    f.barrier()
    r = f.allreduce([rank + 1.0, 10.0 - rank])
    f.cleanup()
    q.put((rank, r))


def test_file_control_plane_two_processes(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    base = str(tmp_path / "ctl")
    ps = [ctx.Process(target=_worker, args=(r, base, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, [2.0, 10.0]), (1, [2.0, 10.0])]
    assert os.listdir(tmp_path) == []


def test_bench_imports_no_torch():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "import torch" not in src
    for f in os.listdir(os.path.join(ROOT, "nx_signal_amd")):
        if f.endswith(".py") and f != "device.py":  # device.py accepts torch tensors handed in by a caller that already uses torch
            assert "import torch" not in open(os.path.join(ROOT, "nx_signal_amd", f)).read(), f
