"""bench.py — the north-star measurement (BASELINE.json): Mrays/s on the 1 003 520-triangle synthetic S1 at
1920x1080, one process per GPU, image tiles sharded across ranks, one RCCL gather of the
packed film.  A "step" = one launch of the hot path = one sample per pixel over the rank's tiles, all bounces
(the reference's vkCmdTraceRaysKHR, offline/main.zig:131-165).  Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 64 --warmup 4
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (before the HIP library: one HIP runtime per process)
import torch.distributed as dist  # noqa: E402

from moonshine_amd import api, scenes  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"


def cpu_baseline(fixture):
    """The oracle (a scalar C restatement, kind="port") timed on this host's cores on a bounded sample of the
    same workload: S1 at 1920x1080, 6 spp (about 10-30 s of CPU work).  Reported baseline only — never the product path."""
    from oracle import orc
    orc.build()
    cores = os.cpu_count() or 1
    c = orc.Context(threads=cores)
    W, H, SPP = 1920, 1080, 16
    s, l = scenes.s1(c, extent=(W, H))
    c.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
    c.render(s, l, launches=1)          # builds the BVH, touches memory
    c.reset_counters()
    t0 = time.perf_counter()
    c.render(s, l, launches=SPP)
    dt = time.perf_counter() - t0
    k = c.counters()
    rays = k["closest_rays"] + k["shadow_rays"]
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "S1 (1 003 520 tris) at %dx%d, %d spp, max_bounces 8, env+mesh NEE; %d rays in %.2f s on %d threads; Msamples/s %.3f"
                      % (W, H, SPP, rays, dt, cores, k["samples"] / dt / 1e6)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--env", default="constant", choices=["constant", "sky"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (a.gpus, world))
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
    sensor, lens = scenes.s1(ctx, extent=(a.width, a.height), env=a.env)
    ctx.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
    ctx.set_profiling(kernel_events=True, traversal_counters=False)
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
    ctx.clear_sensor(sensor)
    ctx.reset_stats()
    sync()
    t0 = time.perf_counter()
    ctx.render(sensor, lens, launches=a.steps, readback=False)   # EXACTLY K steps; returns after the stream is idle
    gather()
    sync()
    dt = time.perf_counter() - t0
    st = ctx.stats()

    tt = torch.tensor([dt, float(st["closest_rays"]), float(st["shadow_rays"]), float(st["samples"])], dtype=torch.float64, device="cuda")
    if world > 1:
        tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
    closest, shadow, samples = float(tt[1]), float(tt[2]), float(tt[3])
    rays = closest + shadow

    if rank == 0:
        fx = json.load(open(os.path.join(ROOT, "tests", "golden", "roofline_s1.json")))
        b_ray = fx["B_ray"]
        # dominant kernel: k_trace_closest — algorithmic bytes per launch / average launch duration (HIP events on the render stream)
        nl = max(int(st["trace_closest_launches"]), 1)
        avg_ms = st["trace_closest_ms"] / nl
        bytes_per_launch = b_ray * st["closest_rays"] / nl
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM bytes per launch from the committed PMC passes (tools/profile_round.sh; FETCH_SIZE x2 + WRITE_SIZE) — only valid
        # for the exact configuration they were collected on
        traffic = None
        tps = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json"))   # newest round last
        if tps and world == 1 and a.steps == 64 and (a.width, a.height, a.env) == (1920, 1080, "constant"):
            traffic = json.load(open(os.path.join(ROOT, "profiles", tps[-1])))["traffic_bytes_per_launch"]
        out = {
            "metric": "Mrays/sec, 1M-tri scene @1080p", "value": rays / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "S1: 7x7 order-5 icospheres (1 003 520 tris) + ground + emissive quad, %dx%d, %d spp, max_bounces 8, env+mesh NEE with MIS, %s env"
                                   % (a.width, a.height, a.steps, a.env),
                       "sharding": "16x16 image tiles, tile t -> rank t mod %d, one RCCL gather of the packed film" % world},
            "msamples_per_s": samples / dt / 1e6,
            "rays": {"closest": closest, "shadow": shadow, "per_sample": rays / max(samples, 1.0)},
            "roofline": {"bound": "hbm", "kernel": "k_trace_closest", "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / (HBM_PEAK / 1e9), "traffic": traffic, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "bytes_per_ray": b_ray, "rays_per_launch": st["closest_rays"] / nl, "avg_launch_ms": avg_ms, "launches": nl},
            "whole_path_roofline_frac": (rays / dt * b_ray + samples / dt * (fx["B_shade"] + 32)) / HBM_PEAK / world,
            "kernel_ms": {"trace_closest": st["trace_closest_ms"], "trace_shadow": st["trace_shadow_ms"], "shade": st["shade_ms"], "render": st["render_ms"]},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(fx)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
