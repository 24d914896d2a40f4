"""Tile sharding bookkeeping: the image is cut into tile_size² tiles; tile t belongs to shard
t mod shard_count; a shard's packed film holds its tiles in order, each tile row-major, padded to the largest shard.
Pure index math (host logic) — the device-side equivalents are shard_pixel()/k_unpack_film in csrc/integrator.hip."""
import numpy as np

# 16x16 = one k_shade workgroup per tile.  Measured on S1 1080p (tools/shard_time.py): against 64x64 tiles the rays per
# rank at 8 shards spread 2.5 % instead of 9.6 % (the job runs at the slowest rank), and the 1-GPU frame is 1 % faster.
DEFAULT_TILE = 16


def tile_grid(width, height, tile_size=DEFAULT_TILE):
    return (width + tile_size - 1) // tile_size, (height + tile_size - 1) // tile_size


def shard_tiles(width, height, shard_index, shard_count, tile_size=DEFAULT_TILE):
    tx, ty = tile_grid(width, height, tile_size)
    return list(range(shard_index, tx * ty, shard_count))


def padded_tiles_per_shard(width, height, shard_count, tile_size=DEFAULT_TILE):
    tx, ty = tile_grid(width, height, tile_size)
    return (tx * ty + shard_count - 1) // shard_count


def pack(film, shard_index, shard_count, tile_size=DEFAULT_TILE):
    """row-major film (H, W, C) -> packed (padded_tiles * tile_size², C) for one shard."""
    h, w, c = film.shape
    tx, _ = tile_grid(w, h, tile_size)
    per = padded_tiles_per_shard(w, h, shard_count, tile_size)
    out = np.zeros((per, tile_size, tile_size, c), film.dtype)
    for k, t in enumerate(shard_tiles(w, h, shard_index, shard_count, tile_size)):
        x0, y0 = (t % tx) * tile_size, (t // tx) * tile_size
        blk = film[y0:y0 + tile_size, x0:x0 + tile_size]
        out[k, :blk.shape[0], :blk.shape[1]] = blk
    return out.reshape(per * tile_size * tile_size, c)


def unpack(gathered, width, height, shard_count, channels=4, tile_size=DEFAULT_TILE):
    """concatenation of `shard_count` packed films (shard order) -> row-major (H, W, C)."""
    tx, _ = tile_grid(width, height, tile_size)
    per = padded_tiles_per_shard(width, height, shard_count, tile_size)
    g = np.asarray(gathered).reshape(shard_count, per, tile_size, tile_size, channels)
    film = np.zeros((height, width, channels), g.dtype)
    for s in range(shard_count):
        for k, t in enumerate(shard_tiles(width, height, s, shard_count, tile_size)):
            x0, y0 = (t % tx) * tile_size, (t // tx) * tile_size
            hh, ww = min(tile_size, height - y0), min(tile_size, width - x0)
            film[y0:y0 + hh, x0:x0 + ww] = g[s, k, :hh, :ww]
    return film
