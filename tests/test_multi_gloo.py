"""The N > 1 path on CPU: world_size-2 gloo processes run the same sharding and
image-gather code bench.py uses with RCCL on GPUs."""
import os
import socket

import numpy as np
import pytest

from wefax_amd.multi import capture_shard


def test_capture_shard_partitions_exactly():
    for n in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in capture_shard(n, world, r)]
            assert got == list(range(n))
            sizes = [len(capture_shard(n, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        capture_shard(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    from wefax_amd.multi import ImageExchange, capture_shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ex = ImageExchange(dist, torch, capacity=5512 * 4 * 9, device="cpu")
        for step in range(3):                            # several steps reuse the same buffers
            mine = list(capture_shard(5, world, rank))   # 5 captures over 2 ranks -> 3 + 2
            h = 3 + rank + step
            w = 5512 if rank == 0 else 2756
            rng = np.random.default_rng(100 * step + rank)
            img = rng.integers(0, 256, size=(4 * h, w), dtype=np.uint8)
            ex.payload_view()[:img.size] = torch.from_numpy(img.reshape(-1))
            got = ex.gather(img.size, w)
            if rank == 0:
                assert len(got) == world
                for r, (buf, wr) in enumerate(got):
                    hr = 3 + r + step
                    ref = np.random.default_rng(100 * step + r).integers(
                        0, 256, size=(4 * hr, 5512 if r == 0 else 2756), dtype=np.uint8)
                    assert wr == ref.shape[1]
                    assert np.array_equal(buf.numpy().reshape(ref.shape), ref)
            else:
                assert got is None
            assert mine == ([0, 1, 2] if rank == 0 else [3, 4])
        # max-over-ranks timing reduction as in bench.py
        t = torch.tensor([0.5 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t) == 0.5 + (world - 1)
        with pytest.raises(ValueError):
            ex.gather(ex.capacity + 1, 1)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_image_gather_world_size_2_gloo(tmp_path):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


@pytest.mark.gpu
def test_pipelined_rccl_gather_single_rank(tmp_path):
    """bench.py's N > 1 path (decode k+1 overlapped with the RCCL gather of image k, wefax_amd/multi.py:
    PipelinedExchange) with ONE forced rank: the gathered image equals the locally fetched one."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WFX_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu", "--short"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                       # the one-JSON-line contract survives RCCL's banner
    out = json.loads(lines[0])
    assert out["rccl_gather_checked"] is True and out["n_gpus"] == 1
