"""CPU tests of the N>1 path: tile sharding + gather over torch.distributed (gloo, world_size 2).
Each rank renders only its tiles (here with the oracle standing in for the GPU), the packed films are gathered to
rank 0 and unpacked; the result must equal the unsharded render bit for bit (RNG is keyed by pixel, main.hlsl:85)."""
import os
import subprocess
import sys

import numpy as np

from moonshine_amd import tiles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pack_unpack_roundtrip():
    rs = np.random.default_rng(0)
    for (w, h, g, ts) in ((200, 136, 3, 64), (64, 64, 2, 64), (130, 70, 8, 32), (50, 20, 4, 64)):
        film = rs.random((h, w, 4)).astype(np.float32)
        parts = [tiles.pack(film, i, g, ts) for i in range(g)]
        assert len({p.shape for p in parts}) == 1
        out = tiles.unpack(np.concatenate(parts), w, h, g, 4, ts)
        assert np.array_equal(out, film)
    assert tiles.shard_tiles(1920, 1080, 3, 8, 64) == list(range(3, 30 * 17, 8))
    assert tiles.shard_tiles(1920, 1080, 3, 8) == list(range(3, 120 * 68, 8))   # default 16x16 tiles


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from oracle import orc
from moonshine_amd import scenes, tiles
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
W, H = 150, 100
TS = tiles.DEFAULT_TILE   # the library's default tile size
c = orc.Context(tile_size=TS, shard_index=rank, shard_count=world, threads=2)
s, l = scenes.cornell(c, extent=(W, H))
c.set_pipeline(samples_per_run=1, max_bounces=4, env_samples_per_bounce=0, mesh_samples_per_bounce=1)
c.render(s, l, launches=2)
mine = torch.from_numpy(tiles.pack(c.sensor_data(s), rank, world, TS))
parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
dist.gather(mine, parts, dst=0)
if rank == 0:
    film = tiles.unpack(torch.cat(parts).numpy(), W, H, world, 4, TS)
    ref = orc.Context(threads=2)
    s2, l2 = scenes.cornell(ref, extent=(W, H))
    ref.set_pipeline(samples_per_run=1, max_bounces=4, env_samples_per_bounce=0, mesh_samples_per_bounce=1)
    ref.render(s2, l2, launches=2)
    assert np.array_equal(film.view(np.uint32), ref.sensor_data(s2).view(np.uint32)), "sharded film differs from unsharded"
    print("SHARDED_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gloo_gather(tmp_path, orc):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    import socket
    with socket.socket() as so:      # a port nobody holds right now (a fixed one makes the rendezvous wait for whoever has it)
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(script), ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "SHARDED_OK" in out.stdout
