"""usage (GPU box): python tools/hull_sweep.py HARSH FAR|none FIRST LAST — tests/hull_rays.py scenes of seeds FIRST..LAST-1 through the HIP path and the oracle, counting the rays whose
hit record or occlusion differs (what tests/test_gpu_parity.py::test_rays_at_the_hulls_of_far_scaled_and_sheared_instances asserts, as a count: profiles/r05_fuzz_sweeps.txt)"""
import sys; sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import numpy as np
from oracle import orc
import moonshine_amd.api as api
import hull_rays
harsh=int(sys.argv[1]); far=None if sys.argv[2]=="none" else float(sys.argv[2]); a,b=int(sys.argv[3]),int(sys.argv[4])
nbad=0; nrays=0; badseeds=[]
for seed in range(a,b):
    oc=orc.Context(threads=1); gc=api.Context()
    world=hull_rays.hull_scene(oc,seed,harsh=bool(harsh)); hull_rays.hull_scene(gc,seed,harsh=bool(harsh))
    for c in (oc,gc): c.create_sensor(8,8)
    rays=hull_rays.hull_rays(world,seed,far=far)
    ids,tuv=gc.trace_rays(rays,any_hit=False); occ,_=gc.trace_rays(rays,any_hit=True)
    k0=nbad
    for k in range(len(rays)):
        hit,oid,otuv=oc.trace_closest(rays[k,:3],rays[k,3:6],float(rays[k,6])); sh=oc.trace_shadow(rays[k,:3],rays[k,3:6],float(rays[k,6]))
        bad = bool(ids[k,0])!=hit or (hit and (tuple(ids[k,1:4])!=tuple(oid) or not np.array_equal(tuv[k].view(np.uint32),otuv.view(np.uint32)))) or bool(occ[k,0])!=sh
        nbad+=bad
    nrays+=len(rays)
    if nbad>k0: badseeds.append(seed)
    del oc,gc
print("harsh",harsh,"far",far,"seeds",a,b,"rays",nrays,"differ",nbad,"in seeds",badseeds[:20])
