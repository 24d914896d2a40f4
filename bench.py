"""bench.py — the north-star measurement (BASELINE.json): Mrays/s on the 1 003 520-triangle synthetic S1 at
1920x1080, one process per GPU, image tiles sharded across ranks, one RCCL gather of the
packed film.  A "step" = one launch of the hot path = one sample per pixel over the rank's tiles, all bounces
(the reference's vkCmdTraceRaysKHR, offline/main.zig:131-165).  Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 64 --warmup 4 [--scene s2] [--env sky]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --gpus N ...                 (no launcher: bench.py starts the N ranks itself as CHILD processes, before it touches the GPU)
    python bench.py --gpus N --launcher group    (one process: the library's own MsneGroup — one context per GPU, native ncclGather, csrc/group.hip)

`transport` in the line says what moved the films: "rccl" (torch.distributed nccl backend or the library's ncclGather between distinct GPUs), "gloo" (ranks that
had to share a GPU: host memory), "copy" (group members sharing a GPU: device-to-device copies), "none" (one rank).  `ranks_seen` / `devices_seen` are counted, not assumed.

Timed region: EXACTLY `steps` steps, bracketed by barrier + synchronize on both sides, max over ranks.  The region is
repeated `--repeats` times (default 5, SURVEY.md §8(d): "median of >= 5 runs") on a cleared sensor; `value` is the median
repeat, every repeat is listed in `repeat_values`.
After the headline (N = 1, S1, constant environment): `other_configs` = the same K-step batch on BASELINE.json's other single-GPU configurations — S2 (configs[4]: 10.24 M
instanced triangles), S1 under the image environment and the configs[2] stand-in (textured, GLB + PIZ EXR through the importers) — each in a context of its own, median of 3
(`--no-other-configs` skips them; never the headline `value`); `--sustain-seconds S` (default 3): the batch back to back for >= S seconds (`sustained`: rate per 1-s window,
shader clock probed while it runs; 0 = off).
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402,F401
import torch  # noqa: E402  (before the HIP library: one HIP runtime per process)
import torch.distributed as dist  # noqa: E402

from moonshine_amd import api, scenes  # noqa: E402
from moonshine_amd.hostinfo import usable_cores, source_hash  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"
KERNELS = ("k_trace_closest", "k_trace_shadow", "k_shade")


_STANDIN = None


def build_scene(ctx, a):
    if a.scene == "s2":
        return scenes.s2(ctx, extent=(a.width, a.height))
    if a.scene == "standin":     # configs[2]'s workload on the substituted asset, through the GLB and EXR importers (GPU contexts only: the oracle is fed by tests/)
        global _STANDIN
        _STANDIN = _STANDIN or standin_files()
        lens, _ = ctx.load_glb(_STANDIN[0]); ctx.set_background_exr(_STANDIN[1])
        return ctx.create_sensor(a.width, a.height), lens
    return scenes.s1(ctx, extent=(a.width, a.height), env=a.env)


def standin_files():
    """the configs[2] stand-in (the Salle-de-bain asset exists nowhere offline: SURVEY.md §8(d) "asset substituted") written to a scratch directory: a GLB of 983 052
    textured triangles in 54 transformed instances, 196 PNG textures, and a 2048x1024 PIZ-compressed HDR environment — the generator the parity tests use
    (tests/io_common.py write_bathroom_standin; about ten seconds of host time, outside every timed region)"""
    import tempfile
    from tests import io_common as io
    d = tempfile.mkdtemp(prefix="msne_standin_")
    import atexit, shutil
    atexit.register(shutil.rmtree, d, ignore_errors=True)      # 26 MB of scratch: gone when the process ends
    glb, exr = os.path.join(d, "bath.glb"), os.path.join(d, "sky.exr")
    io.write_bathroom_standin(glb, exr)
    return glb, exr


def other_config(a, dev, scene, env):
    """BASELINE.json's other single-GPU configuration(s) in the same record (never the headline `value`): the same K-step batch on another scene, median of 3 repeats,
    in a context of its own after the headline measurement is over.  configs[4] = S2, the 10 M-triangle instanced traversal stress; "standin" = configs[2]'s workload on
    the substituted asset (textured k_shade instantiation, image environment with the mip descent, GLB + EXR importers)."""
    b = argparse.Namespace(scene=scene, env=env, width=a.width, height=a.height, steps=a.steps)
    c = api.Context(device=dev)
    sensor, lens = build_scene(c, b)
    name = workload_name(b)
    c.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
    c.render(sensor, lens, launches=0, readback=False)
    c.set_profiling(kernel_events=False, traversal_counters=False)
    c.reserve(sensor, max(a.steps, a.warmup))
    if a.warmup:
        c.render(sensor, lens, launches=a.warmup, readback=False)
    c.reset_stats()
    ts = []
    for _ in range(3):
        c.clear_sensor(sensor); torch.cuda.synchronize()
        t0 = time.perf_counter(); c.render(sensor, lens, launches=a.steps, readback=False); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    st = c.stats(); dt = statistics.median(ts)
    rays = (st["closest_rays"] + st["shadow_rays"]) / 3.0
    c.close()
    return {"workload": name, "value": rays / dt / 1e6, "unit": "Mrays/s", "ms_per_step": dt / a.steps * 1e3, "steps": a.steps, "repeats": 3,
            "msamples_per_s": st["samples"] / 3.0 / dt / 1e6}


def cpu_baseline(a):
    """The oracle (a scalar C restatement, kind="port") timed on this host's cores on a bounded sample of the same
    workload (about 10-30 s of CPU work): all hardware threads at full resolution, and ONE thread at 480x270 (SURVEY.md
    §8(d) asks for both).  Reported baseline only — never the product path."""
    from oracle import orc
    orc.build()
    cores = usable_cores()

    def run(threads, W, H, spp):
        c = orc.Context(threads=threads)
        if a.scene == "standin":     # the oracle reads no files itself: the parity tests' loader feeds it the same GLB and EXR (tests/io_common.py oracle_load)
            global _STANDIN
            from tests import io_common as io
            _STANDIN = _STANDIN or standin_files()
            l, _ = io.oracle_load(orc, c, _STANDIN[0], _STANDIN[1]); s = c.create_sensor(W, H)
        else:
            s, l = build_scene(c, argparse.Namespace(scene=a.scene, env=a.env, width=W, height=H))
        c.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
        c.render(s, l, launches=1)          # builds the BVH, touches memory
        c.reset_counters()
        t0 = time.perf_counter()
        c.render(s, l, launches=spp)
        dt = time.perf_counter() - t0
        k = c.counters()
        return k["closest_rays"] + k["shadow_rays"], k["samples"], dt

    W, H, SPP = a.width, a.height, 16 if cores >= 64 else (4 if cores >= 12 else 1)
    rays, samples, dt = run(cores, W, H, SPP)
    rays1, samples1, dt1 = run(1, 480, 270, 1)
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%s at %dx%d, %d spp, max_bounces 8, env+mesh NEE; %d rays in %.2f s on %d threads; Msamples/s %.3f"
                      % (a.scene.upper(), W, H, SPP, rays, dt, cores, samples / dt / 1e6),
            "one_thread": {"value": rays1 / dt1 / 1e6, "unit": "Mrays/s", "cores": 1,
                           "sample": "%s at 480x270, 1 spp; %d rays in %.2f s on 1 thread" % (a.scene.upper(), rays1, dt1)}}


def committed_counters(scene):
    """profiles/<tag>_counters_<scene>.json of the NEWEST tag that has this scene: per-RAY figures from the committed rocprofv3 PMC passes of this
    command (HBM bytes, VALU instructions by class, lanes per instruction) — per ray, so they apply at any --steps.  Returns (dict, file name)."""
    d = os.path.join(ROOT, "profiles")
    fs = sorted(f for f in os.listdir(d) if f.endswith("_counters_%s.json" % scene))
    return (json.load(open(os.path.join(d, fs[-1]))), fs[-1]) if fs else (None, None)


def valu_calibration():
    """profiles/r03_valu_calibration.json: issue cycles per wave-instruction by class, measured by tools/valu_microbench.hip on this part"""
    f = os.path.join(ROOT, "profiles", "r03_valu_calibration.json")
    return json.load(open(f)) if os.path.exists(f) else None


def l1_tag_calibration():
    """profiles/*_l1_tag_calibration.json (newest): the largest L1 tag-access rate per clock per CU any variant of tools/gather_microbench.hip reached under the counters"""
    d = os.path.join(ROOT, "profiles")
    fs = sorted(f for f in os.listdir(d) if f.endswith("_l1_tag_calibration.json"))
    return (json.load(open(os.path.join(d, fs[-1]))), fs[-1]) if fs else (None, None)


def gpu_count_without_hip():
    """GPUs this process may use, counted WITHOUT bringing up a HIP / HSA runtime in it (the parent of the ranks must not hold one): the visibility masks if set,
    else the KFD topology (a node with SIMDs is a GPU)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    n = 0
    try:
        top = "/sys/class/kfd/kfd/topology/nodes"
        for node in os.listdir(top):
            for line in open(os.path.join(top, node, "properties")):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
    except OSError:
        pass
    return n


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (torch.distributed.run, rendezvous on 127.0.0.1) and relay rank 0's
    line.  This process never touches the GPU (the devices are counted from the visibility masks / the KFD topology, not through HIP) and is never replaced by
    another.  With fewer GPUs than ranks (a one-GPU box) the ranks share devices and gather through host memory (gloo) — labelled as such in the line."""
    import subprocess
    ndev = gpu_count_without_hip()
    env = dict(os.environ)
    if ndev < a.gpus and "MSNE_BENCH_BACKEND" not in env:
        env["MSNE_BENCH_BACKEND"] = "gloo"
    # --standalone: torch.distributed.run picks and HOLDS a free rendezvous port itself (no bind-then-close race on a busy node)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in p.stdout:
        sys.stdout.write(line); sys.stdout.flush()
    raise SystemExit(p.wait())


def run_group(a):
    """--launcher group: the library's own multi-GPU driver (csrc/group.hip) — one context per GPU inside THIS process, one host thread per member, tiles sharded
    over the members, ncclGather of the packed films (device copies when members share a GPU), k_unpack_film on member 0.  Timed region: MsneGroupRender of
    exactly K steps (render on every member + gather + unpack; it returns when the assembled film is complete), bracketed by device synchronisation."""
    ndev = max(torch.cuda.device_count(), 1)
    devices = [i % ndev for i in range(a.gpus)]
    g = api.Group(devices)
    sensor, lens = g.build(lambda c: build_scene(c, a))
    g.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
    for c in g.members:
        c.set_profiling(kernel_events=False, traversal_counters=False)
        c.reserve(sensor, max(a.steps, a.warmup))

    def sync():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)
    if a.warmup:
        g.render(sensor, lens, launches=a.warmup)
    for c in g.members:
        c.reset_stats()
    times = []
    for _ in range(max(a.repeats, 1)):
        for c in g.members:
            c.clear_sensor(sensor)
        sync()
        t0 = time.perf_counter()
        g.render(sensor, lens, launches=a.steps)
        sync()
        times.append(time.perf_counter() - t0)
    R = len(times)
    st = [c.stats() for c in g.members]
    closest, shadow, samples = (sum(float(s[k]) for s in st) / R for k in ("closest_rays", "shadow_rays", "samples"))
    rays, dt = closest + shadow, statistics.median(times)
    if a.dump_film:
        np.save(a.dump_film, g.sensor_data(sensor))
    rates = [rays / t / 1e6 for t in times]
    print(json.dumps({
        "metric": "Mrays/sec, 1M-tri scene @1080p" if a.scene == "s1" else ("Mrays/sec, 10M-tri instanced scene @1080p" if a.scene == "s2" else "Mrays/sec, 1M-tri textured interior (asset substituted) @1080p"),
        "value": rays / dt / 1e6, "unit": "Mrays/s", "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload_name(a), "sharding": "16x16 image tiles, tile t -> member t mod %d, one gather of the packed film (MsneGroup, one process)" % a.gpus},
        "launcher": "group", "transport": g.transport(), "ranks_seen": len(g.members), "devices_seen": len(set(devices)),
        "repeats": R, "repeat_values": rates, "spread": (max(rates) - min(rates)) / statistics.median(rates),
        "msamples_per_s": samples / dt / 1e6, "rays": {"closest": closest, "shadow": shadow, "per_sample": rays / max(samples, 1.0)},
        "roofline": None, "cpu_baseline": None}))
    g.close()


def workload_name(a):
    if a.scene == "standin":
        return ("configs[2] stand-in (asset substituted): 983 052 textured triangles in 54 instances, 196 PNG textures, 2048x1024 PIZ HDR environment"
                ", %dx%d, %d spp, max_bounces 8, env+mesh NEE with MIS" % (a.width, a.height, a.steps))
    return (("S1: 7x7 order-5 icospheres (1 003 520 tris) + ground + emissive quad" if a.scene == "s1" else
             "S2: 500 instances of one order-5 icosphere (10 240 000 instanced tris), glass / GGX")
            + ", %dx%d, %d spp, max_bounces 8, env+mesh NEE with MIS, %s env" % (a.width, a.height, a.steps, a.env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launcher", default="auto", choices=["auto", "ranks", "group"],
                    help="how --gpus N > 1 runs when no launcher started us: ranks = child processes under torch.distributed.run (auto), group = MsneGroup in this process")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="s1", choices=["s1", "s2", "standin"])
    ap.add_argument("--env", default="constant", choices=["constant", "sky"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the secondary single-GPU configurations (S2, S1 under the sky, the configs[2] stand-in) reported next to the headline")
    ap.add_argument("--sustain-seconds", type=float, default=3.0,
                    help="after the timed repeats: back-to-back K-step batches for at least this long, reported as `sustained` (rate per 1-s window, shader clock at both ends); 0 = off")
    ap.add_argument("--no-gpu-visits", action="store_true", help="skip the extra pass that counts the GPU's own node visits / triangle tests per ray (profiling runs: its kernels would be counted)")
    ap.add_argument("--dump-film", default=None, help="rank 0 saves the assembled film of the last repeat here (.npy): parity tests of the gather path")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launched = "WORLD_SIZE" in os.environ or "RANK" in os.environ
    if a.gpus > 1 and not launched:
        return run_group(a) if a.launcher == "group" else launch_ranks(a)
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    # MSNE_BENCH_BACKEND=gloo is a debugging aid: several ranks may then share one GPU and the gather goes through host memory
    backend = os.environ.get("MSNE_BENCH_BACKEND", "nccl")
    dev = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)

    ctx = api.Context(device=dev, shard_index=rank, shard_count=world)
    t_scene = time.perf_counter()
    sensor, lens = build_scene(ctx, a)
    ctx.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
    t_build = time.perf_counter()
    ctx.render(sensor, lens, launches=0, readback=False)      # scene -> first launch: texture / material upload, BLAS + TLAS build, light tables (outside the timed region)
    torch.cuda.synchronize()
    build_ms = {"scene_calls_ms": (t_build - t_scene) * 1e3, "build_ms": (time.perf_counter() - t_build) * 1e3}
    ctx.set_profiling(kernel_events=False, traversal_counters=False)   # no events inside the timed region: per-kernel times come from the attribution pass below
    ctx.reserve(sensor, max(a.steps, a.warmup))   # wavefront state for the whole batch, allocated outside the timed region
    ptr, n4 = ctx.packed_film(sensor)
    gathered = torch.empty(world * n4 * 4, dtype=torch.float32, device="cuda") if rank == 0 else None

    def film_tensor():
        # zero-copy view of the library's packed film as a torch tensor (plumbing for the RCCL gather)
        class _W:
            __cuda_array_interface__ = {"shape": (n4 * 4,), "typestr": "<f4", "data": (ptr, False), "version": 2}
        return torch.as_tensor(_W(), device="cuda")

    def gather():
        # ONE gather of the packed films to rank 0 (RCCL over xGMI), then k_unpack_film on rank 0
        if world == 1:
            return
        t = film_tensor()
        if backend != "nccl":
            t = t.cpu()
        if rank == 0:
            parts = list(gathered.view(world, -1).unbind(0)) if backend == "nccl" else [torch.empty_like(t) for _ in range(world)]
            dist.gather(t, parts, dst=0)
            if backend != "nccl":
                gathered.copy_(torch.cat(parts))
            torch.cuda.synchronize()
            ctx.unpack_gathered(sensor, gathered.data_ptr(), world)
        else:
            dist.gather(t, None, dst=0)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if a.warmup:
        ctx.render(sensor, lens, launches=a.warmup, readback=False)
    gather()
    warm = ctx.stats()
    ctx.reset_stats()
    times, t_render, t_gather = [], [], []
    for _ in range(max(a.repeats, 1)):
        ctx.clear_sensor(sensor)
        sync()
        t0 = time.perf_counter()
        ctx.render(sensor, lens, launches=a.steps, readback=False)   # EXACTLY K steps; returns after the stream is idle
        t1 = time.perf_counter()
        gather()
        t2 = time.perf_counter()
        sync()
        times.append(time.perf_counter() - t0)
        t_render.append(t1 - t0); t_gather.append(t2 - t1)           # this rank's own share: its K steps, then the gather (a sender returns once its film is handed over; rank 0 also unpacks)
    st = ctx.stats()          # sums over the repeats (every repeat traces the same rays: same sample indices)
    R = len(times)
    # attribution pass (outside the timed region): the same K steps once more with the kernels in STREAM ORDER, so that each kernel's HIP-event
    # duration is exclusive (in the timed repeats k_trace_shadow(b) overlaps k_trace_closest(b+1) on a second stream and both durations include
    # the other's share of the SIMDs).  The roofline's kernel and its launch durations come from this pass: they do not depend on --steps.
    ctx.reset_stats(); ctx.set_profiling(kernel_events=2, traversal_counters=False)
    ctx.clear_sensor(sensor); sync()
    ctx.render(sensor, lens, launches=a.steps, readback=False)
    sync()
    sa = ctx.stats()
    # ... and what each bounce of this rank's shard cost in that pass: launches arrive as closest(b), shade(b), shadow(b) per batch (moonshine_amd.h MsneGetLaunchTimes)
    NB = 8 + 2                                                         # max_bounces + 2 passes per batch (HdMoonshine::render max_iter)
    per_bounce = {0: [0.0] * NB, 1: [0.0] * NB, 2: [0.0] * NB}; seen = {0: 0, 1: 0, 2: 0}
    for kind, ms in ctx.launch_times():
        per_bounce[kind][seen[kind] % (NB - 1 if kind == 1 else NB)] += ms; seen[kind] += 1
    ctx.set_profiling(kernel_events=False, traversal_counters=False)

    # ... and once more with the traversal kernels' COUNTING instantiations (outside the timed region, never part of `value`): the GPU's own node visits and triangle tests
    # per ray in THIS run — a quantity of the roofline object that is observed here and not read from a committed file (`roofline.gpu_visits`).  Plain device
    # synchronisation only: a rank that fails here must not leave the others in a barrier.
    gpu_visits = None
    if not a.no_gpu_visits:
        try:
            ctx.reset_stats(); ctx.set_profiling(kernel_events=False, traversal_counters=True)
            ctx.clear_sensor(sensor); torch.cuda.synchronize()
            ctx.render(sensor, lens, launches=a.steps, readback=False); torch.cuda.synchronize()
            sv = ctx.stats(); tc = ctx.traversal_counters()
            cr, sr = max(float(sv["closest_rays"]), 1.0), max(float(sv["shadow_rays"]), 1.0)
            gpu_visits = {"node_visits_per_closest_ray": tc["closest_node_visits"] / cr, "triangle_tests_per_closest_ray": tc["closest_tri_tests"] / cr,
                          "node_visits_per_shadow_ray": tc["shadow_node_visits"] / sr, "triangle_tests_per_shadow_ray": tc["shadow_tri_tests"] / sr,
                          "bytes_per_closest_ray": tc["closest_node_visits"] / cr * 80.0 + tc["closest_tri_tests"] / cr * 48.0 + 48.0,
                          "from": "this run: one extra pass of the same %d steps with the traversal kernels' counting instantiations (MsneSetProfiling traversal_counters)" % a.steps}
        except Exception as e:      # (a diagnostic must not cost the line)
            gpu_visits = {"error": str(e)}
        ctx.set_profiling(kernel_events=False, traversal_counters=False)
        ctx.reset_stats()

    # A render is seconds of launches (offline/main.zig:131-165), the timed region above a burst of tens of milliseconds: with --sustain-seconds S the same K-step batch
    # runs back to back for at least S seconds.  Per-batch wall times give the rate per 1-s window; the shader clock is probed by one wave on a stream of its own
    # (MsneProbeClockGhz) from a second host thread while the batches run.
    sustained = None
    if a.sustain_seconds > 0:
        import threading
        clocks, stop = [], threading.Event()

        def probe():
            while not stop.is_set():
                g = api.load_library().MsneProbeClockGhz(dev)
                if g > 0:
                    clocks.append((time.perf_counter(), g))
                stop.wait(0.25)
        ctx.reset_stats(); sync()
        th = threading.Thread(target=probe, daemon=True); th.start()
        marks = [time.perf_counter()]
        while True:
            # how long to go on is rank 0's call, told to everybody: every batch ends in a collective (the gather), and ranks that counted batches by their own clocks
            # would sooner or later disagree by one — and wait for each other for ever (round 6: the two-rank test hung exactly so)
            go = torch.tensor([1.0 if marks[-1] - marks[0] < a.sustain_seconds else 0.0], dtype=torch.float64, device="cuda")
            if world > 1:
                dist.broadcast(go, src=0)
            if float(go[0]) == 0.0:
                break
            ctx.clear_sensor(sensor)
            ctx.render(sensor, lens, launches=a.steps, readback=False)
            gather()
            marks.append(time.perf_counter())
        sync()
        stop.set(); th.join()
        ss = ctx.stats(); nb = max(len(marks) - 1, 1)
        rays_all = torch.tensor([float(ss["closest_rays"] + ss["shadow_rays"])], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(rays_all, op=dist.ReduceOp.SUM)      # the whole job's rays, not this rank's share
        rays_batch = float(rays_all[0]) / float(nb)
        total_s = marks[-1] - marks[0]
        wins, w0, k0 = [], marks[0], 0      # rate of every whole 1-s window: batches completed in it / its length
        for k in range(1, len(marks)):
            if marks[k] - w0 >= 1.0:
                wins.append(rays_batch * (k - k0) / (marks[k] - w0) / 1e6); w0, k0 = marks[k], k
        third = max(len(clocks) // 3, 1)
        sustained = {"mrays_per_s": rays_batch * nb / total_s / 1e6, "seconds": total_s, "batches": nb, "steps_per_batch": a.steps,
                     "min_1s": min(wins) if wins else None, "median_1s": statistics.median(wins) if wins else None, "max_1s": max(wins) if wins else None,
                     "clock_ghz_first": statistics.median([g for _, g in clocks[:third]]) if clocks else None,
                     "clock_ghz_last": statistics.median([g for _, g in clocks[-third:]]) if clocks else None,
                     "clock_ghz_min": min(g for _, g in clocks) if clocks else None, "clock_probes": len(clocks),
                     "clock_from": "s_memtime / s_memrealtime (100 MHz) around a 0.2-ms spin of one probe wave, while the batches run"}

    NDEV = 64
    tt = torch.tensor(times + [float(st["closest_rays"]) / R, float(st["shadow_rays"]) / R, float(st["samples"]) / R, 1.0] + [float(i == dev) for i in range(NDEV)],
                      dtype=torch.float64, device="cuda")
    if world > 1:
        tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        times = [float(x) for x in tmax[:R]]
        devices_seen = int(tmax[R + 4:].sum())
    else:
        devices_seen = 1
    # every rank's own numbers, so that a scaling result below the prediction can be attributed from the one line: its K steps (median repeat), its gather, the
    # attribution pass by bounce — the thin late bounces of a small shard are where DESIGN.md section 7 expects the loss
    mine = torch.tensor([statistics.median(t_render) * 1e3, statistics.median(t_gather) * 1e3, sa["render_ms"], float(st["closest_rays"] + st["shadow_rays"]) / R]
                        + per_bounce[0] + per_bounce[2] + per_bounce[1], dtype=torch.float64, device="cuda")
    if world > 1:
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
    else:
        allr = [mine]
    per_rank = [{"rank": r, "render_ms": float(v[0]), "gather_ms": float(v[1]), "render_ms_stream_order": float(v[2]), "rays": float(v[3]),
                 "closest_ms_by_bounce": [round(float(x), 4) for x in v[4:4 + NB]], "shade_ms_by_bounce": [round(float(x), 4) for x in v[4 + NB:4 + 2 * NB]],
                 "shadow_ms_by_bounce": [round(float(x), 4) for x in v[4 + 2 * NB:4 + 3 * NB - 1]]} for r, v in enumerate(allr)]
    closest, shadow, samples = float(tt[R]), float(tt[R + 1]), float(tt[R + 2])     # per repeat, all ranks
    ranks_seen = int(round(float(tt[R + 3])))                                         # ranks that took part in the reduction (counted, not assumed)
    rays = closest + shadow
    dt = statistics.median(times)

    observed_backend = dist.get_backend() if world > 1 else "none"
    if rank == 0 and a.dump_film:
        if world == 1:
            ctx.render(sensor, lens, launches=0, readback=True)     # the gather already unpacked the film on rank 0 when world > 1
        np.save(a.dump_film, ctx.sensor_data(sensor))
    if rank == 0:
        fx = json.load(open(os.path.join(ROOT, "tests", "golden", "roofline_%s.json" % a.scene)))
        # bytes per unit of each kernel: V_n * 80 + V_t * 48 + 48 with the kernel's OWN canonical visit counts;
        # k_shade: B_hit per surface hit + the 288 B of wavefront state it reads and writes per path
        unit_bytes = {"k_trace_closest": fx["V_n_closest"] * 80 + fx["V_t_closest"] * 48 + 48,
                      "k_trace_shadow": fx["V_n_shadow"] * 80 + fx["V_t_shadow"] * 48 + 48,
                      "k_shade": fx["B_hit"] + 288.0}
        # per-kernel time, launches and units of THIS rank in the attribution pass (HIP events on the stream each kernel is launched on, kernels in stream order)
        kms = {"k_trace_closest": sa["trace_closest_ms"], "k_trace_shadow": sa["trace_shadow_ms"], "k_shade": sa["shade_ms"]}
        kln = {"k_trace_closest": sa["trace_closest_launches"], "k_trace_shadow": sa["trace_shadow_launches"], "k_shade": sa["shade_launches"]}
        kun = {"k_trace_closest": float(sa["closest_rays"]), "k_trace_shadow": float(sa["shadow_rays"]), "k_shade": float(sa["closest_rays"])}
        dom = max(KERNELS, key=lambda k: kms[k])          # the dominant kernel: most exclusive time
        nl = max(int(kln[dom]), 1)
        avg_ms = kms[dom] / nl
        bytes_per_launch = unit_bytes[dom] * kun[dom] / nl
        hbm_alg = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0       # algorithmic bytes / launch duration
        cnt, cnt_file = committed_counters(a.scene if a.env == "constant" else a.scene + "_sky")
        cal = valu_calibration()
        ck = cnt["kernels"].get(dom) if cnt else None

        def valu_cycles(k, low=False):
            """issue cycles one unit (ray / path) of kernel k needs on a SIMD: its wave-instructions per unit by class (committed PMC pass) x the calibrated
            cost of the class (nominal 2 / 4 / 8 cycles; INT32 at 2).  The instructions NO class counter books ("OTHER": selects, compares, min / max at 4 cycles, but
            also and / or / xor / shifts right / moves at 2 — profiles/r03_valu_classes.txt) are priced at 4, or with low=True at 4 - 2 x the two-cycle share counted
            in the kernels' ISA."""
            e = cnt["kernels"][k]; cc = cal["class_cycles"]
            if "valu_class_per_unit" not in e:
                return None
            cl = e["valu_class_per_unit"]
            listed = sum(cl[c] for c in ("FMA_F32", "MUL_F32", "ADD_F32", "INT32", "CVT", "TRANS_F32"))
            other = cc["OTHER"] - (2.0 * cal.get("other_two_cycle_share", {}).get("value", 0.0) if low else 0.0)
            return sum(cl[c] * cc[c] for c in ("FMA_F32", "MUL_F32", "ADD_F32", "INT32", "CVT", "TRANS_F32")) + (e["valu_wave_instructions_per_unit"] - listed) * other

        hbm = {"achieved": hbm_alg, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": hbm_alg / (HBM_PEAK / 1e9), "algorithmic_bytes_per_launch": bytes_per_launch,
               "bytes_per_unit": unit_bytes[dom]}
        vc = valu_cycles(dom) if (cnt and cal and ck) else None
        valu_frac = vc * (kun[dom] / nl) / (avg_ms * 1e-3) / cal["peak_simd_cycles_per_s"] if (vc is not None and avg_ms > 0) else None
        # the bound is the resource the kernel is closest to: vector-instruction issue when the §8(d) HBM figure is not a bound at all (> 1) or the lower of the two
        if valu_frac is not None and (hbm["frac"] > 1.0 or valu_frac >= hbm["frac"]):
            # the §8(d) HBM figure exceeds the peak: most algorithmic bytes are re-reads that L2 / Infinity Cache serve, HBM is not what bounds this kernel.
            # What does: vector-instruction ISSUE.  achieved = issue cycles the kernel's instructions need per second (calibrated per class on this part),
            # peak = 1024 SIMDs x 2.4 GHz.
            # peak at the clock the kernel ran at when its counters were taken (GRBM_GUI_ACTIVE / duration in the counters file), not the nominal 2.4 GHz
            clock = (ck.get("memory_pipeline") or {}).get("clock_ghz") or 2.4
            peak = 1024.0 * clock * 1e9
            ach = vc * (kun[dom] / nl) / (avg_ms * 1e-3)
            lanes = ck["lanes_per_valu_instruction"]
            ach_low = valu_cycles(dom, low=True) * (kun[dom] / nl) / (avg_ms * 1e-3)
            # `frac`: the unclassified instructions priced by their measured two-cycle share (the best estimate); `frac_upper`: all of them at 4 cycles — an upper estimate
            # that can exceed 1 (S2: 1.08), which says the pricing is too high, not that the kernel beats the machine
            roof = {"bound": "valu", "kernel": dom, "achieved": ach_low / 1e9, "peak": peak / 1e9, "unit": "G SIMD-cycles/s", "frac": ach_low / peak,
                    "frac_upper": ach / peak, "issue_cycles_per_unit": valu_cycles(dom, low=True), "issue_cycles_per_unit_upper": vc,
                    "wave_instructions_per_unit": ck["valu_wave_instructions_per_unit"],
                    "lanes_per_instruction": lanes, "lane_utilisation": lanes / 64.0, "useful_issue_frac": ach_low / peak * lanes / 64.0,
                    "peak_clock_ghz": clock, "class_per_unit": ck["valu_class_per_unit"],
                    "from": {"instruction_counts": "profiles/" + cnt_file, "cycles_per_class": "profiles/r03_valu_calibration.json"},
                    "hbm_algorithmic_frac": hbm["frac"], "hbm_algorithmic": hbm}
        else:
            roof = dict(hbm, bound="hbm", kernel=dom)
        # the other candidate the round-5 verdict named: the L1's tag rate.  Tag accesses per unit (committed PMC pass) x units per launch / (launch duration x the
        # clock the counters file records x 256 CUs), against the largest rate the gather microbenchmark reached (profiles/*_l1_tag_calibration.json).  `bound` names
        # whichever resource the kernel uses the larger fraction of.
        tcal, tcal_file = l1_tag_calibration()
        mp = (ck or {}).get("memory_pipeline") or {}
        if tcal and mp.get("l1_tag_accesses_per_unit") and avg_ms > 0:
            clk = mp.get("clock_ghz") or 2.4
            per_clk = mp["l1_tag_accesses_per_unit"] * (kun[dom] / nl) / (avg_ms * 1e-3) / (clk * 1e9) / 256.0
            roof["l1_tag"] = {"accesses_per_clk_per_cu": per_clk, "peak": tcal["peak_l1_tag_accesses_per_clk_per_cu"], "frac": per_clk / tcal["peak_l1_tag_accesses_per_clk_per_cu"],
                              "accesses_per_unit": mp["l1_tag_accesses_per_unit"], "from": {"accesses": "profiles/" + cnt_file, "peak": "profiles/" + tcal_file}}
            if roof.get("bound") == "valu" and roof["l1_tag"]["frac"] > roof["frac"]:
                roof["bound"] = "l1_tag"
        # which fields of this object are measured in THIS run and which come from committed profiles of the same command
        roof["measured_in_this_run"] = ["kernel", "avg_launch_ms", "launches", "units_per_launch", "achieved (its time base)", "hbm_algorithmic*"]
        roof["from_committed_counters"] = ["issue_cycles_per_unit*", "wave_instructions_per_unit", "class_per_unit", "lanes_per_instruction", "traffic (bytes per unit)", "memory_pipeline", "l1_tag.accesses_per_unit", "peak_clock_ghz"]
        if cnt:
            roof["counters_source_hash"] = cnt.get("source_hash"); roof["counters_steps"] = cnt.get("steps")
            roof["counters_stale"] = cnt.get("source_hash") != source_hash()       # per-ray instruction counts were taken on other kernel sources than the ones running now
        roof.update({"units_per_launch": kun[dom] / nl, "avg_launch_ms": avg_ms, "launches": nl, "traffic": None,
                     "timing": "HIP events around every launch of the kernel in an extra pass of the same K steps with the kernels in stream order (exclusive durations)"})
        if gpu_visits is not None:
            roof["gpu_visits"] = gpu_visits
            roof["measured_in_this_run"].append("gpu_visits")
        if ck:
            roof["traffic"] = ck["hbm_bytes_per_unit"] * kun[dom] / nl           # HBM bytes per launch: PMC figure per unit (committed) x this run's units
            roof["traffic_from"] = "profiles/" + cnt_file + " (FETCH_SIZE x2 + WRITE_SIZE per unit, measured by rocprofv3 on this command) x this run's units per launch"
            if avg_ms > 0:
                roof["traffic_frac"] = roof["traffic"] / (avg_ms * 1e-3) / HBM_PEAK
            if "memory_pipeline" in ck:
                roof["memory_pipeline"] = dict(ck["memory_pipeline"], **{"from": "profiles/" + cnt_file})
        rates = [rays / t / 1e6 for t in times]
        out = {
            "metric": "Mrays/sec, 1M-tri scene @1080p" if a.scene == "s1" else ("Mrays/sec, 10M-tri instanced scene @1080p" if a.scene == "s2" else "Mrays/sec, 1M-tri textured interior (asset substituted) @1080p"),
            "value": rays / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_name(a),
                       "sharding": "16x16 image tiles, tile t -> rank t mod %d, one gather of the packed film (one process per GPU)" % world},
            "launcher": "ranks", "transport": "none" if world == 1 else ("rccl" if observed_backend == "nccl" else observed_backend), "ranks_seen": ranks_seen, "devices_seen": devices_seen,
            # what torch.distributed says it ran the gather on (not the environment variable), and whether every rank had a GPU of its own
            "transport_observed": None if world == 1 else {"backend": observed_backend, "one_gpu_per_rank": devices_seen == world},
            "repeats": R, "repeat_values": rates, "spread": (max(rates) - min(rates)) / statistics.median(rates),
            "msamples_per_s": samples / dt / 1e6,
            "rays": {"closest": closest, "shadow": shadow, "per_sample": rays / max(samples, 1.0)},
            "roofline": roof, "build": build_ms,
            # per rank: wall time of its own K steps and of its gather in the median repeat; the stream-order pass by bounce (sums over the batch's launches of that bounce)
            "per_rank": per_rank,
            "whole_path_roofline_frac": (rays / dt * fx["B_ray"] + samples / dt * (fx["B_shade"] + 32)) / HBM_PEAK / world,
            "kernel_ms_stream_order": {k: kms[k] for k in KERNELS}, "render_ms_stream_order": sa["render_ms"], "render_ms_per_repeat": st["render_ms"] / R,
            "kernel_hbm_algorithmic_frac": {k: unit_bytes[k] * kun[k] / max(kms[k] * 1e-3, 1e-12) / HBM_PEAK for k in KERNELS},
            "kernel_valu_frac": ({k: valu_cycles(k) * kun[k] / max(kms[k] * 1e-3, 1e-12) / (1024.0e9 * ((cnt["kernels"][k].get("memory_pipeline") or {}).get("clock_ghz") or 2.4)) for k in KERNELS}
                                 if (cnt and cal and all("valu_class_per_unit" in cnt["kernels"].get(k, {}) for k in KERNELS)) else None),
            # for tools/profile_counters.py: what the counters of a rocprofv3 run of this command have to be divided by
            "profile_totals": {"closest_rays": warm["closest_rays"] + st["closest_rays"] + sa["closest_rays"], "shadow_rays": warm["shadow_rays"] + st["shadow_rays"] + sa["shadow_rays"],
                               "samples": warm["samples"] + st["samples"] + sa["samples"]},
        }
        if sustained is not None:
            out["sustained"] = sustained     # rank 0's clock over the whole job's rays (at N > 1 every rank runs the same number of batches, gather included)
        if world == 1 and a.scene == "s1" and a.env == "constant" and not a.no_other_configs:
            # the other single-GPU configurations of BASELINE.json, measured after the headline (a second or two): configs[4] (S2) and S1 under the image environment
            out["other_configs"] = {"s2": other_config(a, dev, "s2", "constant"), "s1_sky": other_config(a, dev, "s1", "sky"), "standin": other_config(a, dev, "standin", "image")}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
