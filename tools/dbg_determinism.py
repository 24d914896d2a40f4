"""Debug aid: renders S1 twice in independent contexts (two BVH builds) and reports differing pixels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
from moonshine_amd import api, scenes

w, h, spp = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080, 3)
films = []
for k in range(int(os.environ.get("N", "3"))):
    c = api.Context()
    s, l = scenes.s1(c, extent=(w, h))
    c.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
    c.render(s, l, launches=spp)
    films.append(c.sensor_data(s).copy())
    print("ctx", k, c.counters())
    if k:
        d = (films[0].view(np.uint32) != films[k].view(np.uint32)).any(axis=2)
        ys, xs = np.nonzero(d)
        print("ctx %d vs 0: %d differing pixels" % (k, len(ys)), list(zip(xs[:10].tolist(), ys[:10].tolist())))
        for x, y in list(zip(xs[:5], ys[:5])):
            print("  ", x, y, films[0][y, x], films[k][y, x])
