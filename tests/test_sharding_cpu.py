"""The N > 1 path on CPU: two gloo ranks shard the frames of a step by strips of four 8x8 tiles exactly
as bench.py does (strip_id % world == rank), send their tile-major shards of all views of the step --
4-byte packed pixels, like nrf_quantize_rgbd8's -- with ONE dist.gather to rank 0, and rank 0 untiles.
The GPU kernels are replaced by a per-pixel function, so this covers the host-side partition, padding
and gather logic (the device untile kernel is checked against the same mapping on the GPU)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, q):
    sys.path[:0] = [str(ROOT / "nerf-cuda_amd")]
    import torch
    import torch.distributed as dist

    import nerfhip as nh

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tps = nh.tiles_per_shard(W, H, world)
    tiles = nh.shard_tile_ids(W, H, rank, world)
    tiles_x = (W + 7) // 8
    V = 3  # views per step
    shard = torch.zeros((V, tps * 64), dtype=torch.int32)
    for v in range(V):
        for k, (tx, ty) in enumerate(tiles):  # "render": packed pixel = f(x, y, view), zeros outside the image
            for l in range(64):
                px, py = tx * 8 + (l & 7), ty * 8 + (l >> 3)
                if px < W and py < H:
                    shard[v, k * 64 + l] = px | (py << 10) | ((v + 1) << 20)
    gathered = torch.empty((world, V, tps * 64), dtype=torch.int32) if rank == 0 else None
    dist.gather(shard, [gathered[r] for r in range(world)] if rank == 0 else None, dst=0)
    in_image = [t for t in tiles if t[0] < tiles_x and t[1] < (H + 7) // 8]
    samples = torch.tensor([len(in_image) * 10], dtype=torch.int64)
    dist.all_reduce(samples)
    if rank == 0:
        g = gathered.numpy()
        q.put((np.stack([nh.untile_numpy(g[:, v, :, None], W, H)[..., 0] for v in range(V)]), int(samples.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("W,H", [(40, 24), (37, 19)])
def test_two_rank_tile_sharding_gloo(W, H):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, W, H, q)) for r in range(2)]
    for p in procs:
        p.start()
    img, samples = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ys, xs = np.mgrid[0:H, 0:W]
    assert img.shape == (3, H, W)
    for v in range(3):
        np.testing.assert_array_equal(img[v], xs | (ys << 10) | ((v + 1) << 20))
    total_tiles = ((W + 7) // 8) * ((H + 7) // 8)
    assert samples == total_tiles * 10  # every tile owned by exactly one rank


def test_partition_is_exact_cover():
    sys.path[:0] = [str(ROOT / "nerf-cuda_amd")]
    import nerfhip as nh
    for W, H, world in [(1920, 1080, 8), (1920, 1080, 3), (20, 12, 4), (8, 8, 2), (100, 30, 5)]:
        tiles_x, tiles_y = (W + 7) // 8, (H + 7) // 8
        ids = sorted(t for r in range(world) for t in nh.shard_tile_ids(W, H, r, world) if t[0] < tiles_x)
        assert ids == sorted((x, y) for y in range(tiles_y) for x in range(tiles_x))  # exact cover, no tile twice
        assert max(len(nh.shard_tile_ids(W, H, r, world)) for r in range(world)) == nh.tiles_per_shard(W, H, world)
