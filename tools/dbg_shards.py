"""Debug aid: S1 rendered unsharded and as G tile shards (gathered by hand), reports differing pixels."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
from moonshine_amd import api, scenes

w, h, spp, G = 1920, 1080, 3, int(os.environ.get("G", "2"))
c = api.Context()
s, l = scenes.s1(c, extent=(w, h))
c.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
c.render(s, l, launches=spp)
film = c.sensor_data(s).copy()
print("full", c.counters())
shards = [api.Context(shard_index=i, shard_count=G) for i in range(G)]
hs = []
for sc in shards:
    ss, ll = scenes.s1(sc, extent=(w, h))
    sc.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
    sc.render(ss, ll, launches=spp, readback=False)
    hs.append(ss)
    print("shard", sc.counters())
hip = C.CDLL("libamdhip64.so.7")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
n4 = shards[0].packed_film(hs[0])[1]
gathered = C.c_void_p()
assert hip.hipMalloc(C.byref(gathered), G * n4 * 16) == 0
for i, (sc, ss) in enumerate(zip(shards, hs)):
    assert hip.hipMemcpy(C.c_void_p(gathered.value + i * n4 * 16), C.c_void_p(sc.packed_film(ss)[0]), n4 * 16, 3) == 0
shards[0].unpack_gathered(hs[0], gathered.value, G)
f2 = shards[0].sensor_data(hs[0])
d = (film.view(np.uint32) != f2.view(np.uint32)).any(axis=2)
ys, xs = np.nonzero(d)
print("%d differing pixels" % len(ys), list(zip(xs[:20].tolist(), ys[:20].tolist())))
for x, y in list(zip(xs[:8], ys[:8])):
    print("  ", x, y, film[y, x], f2[y, x])
