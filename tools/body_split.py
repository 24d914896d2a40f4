"""Per-body split of the traversal loop: static VALU instruction counts of the node / triangle / space bodies of the production trace kernels (basic blocks of the
compiler's assembly, see static_split) x how often each body runs and with how many lanes (STATS instantiation, MsneGetTraversalLaneUse) -> wave instructions
per body per ray, next to the total the PMC pass measured (SQ_INSTS_VALU per ray) and to the wave-cycle laps of the STATS build (trace.hip lap(0) / (1) / (3) / (4)).
usage (GPU box):  python tools/body_split.py [w h spp]   (S1, S1 sky, S2)
       (anywhere): python tools/body_split.py --static   (the static table only)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)

SECTIONS = ("node", "tri", "space", "refill")


def static_split(extra_flags=()):
    """{kernel: {body: VALU instructions}} for the four production (non-STATS) trace kernels, from the basic blocks of the compiler's assembly:
    node  = the block that converts the 48 quantised plane bytes (v_cvt_f32_ubyte) — step_node, group_take and the votes in front of it are scheduled into it;
    tri   = from the block that issues the triangle record's 3 x 16-B loads to the loop's back edge (step_tri and the end-of-iteration bookkeeping), without the
            out-of-line f64 fallback and tie blocks;
    space = (two-level scenes) the block with the TLAS leaf's 4 x 16-B loads and the three IEEE divisions of the shear constants;
    refill = the block(s) that load a new ray and compute its constants (lane_begin: three IEEE divisions).
    What is left of the measured instructions per ray is the loop's own bookkeeping: pops, votes, queue and tail handling."""
    src = os.path.join(ROOT, "moonshine_amd", "csrc", "trace.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "trace.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
                               "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-S", "--cuda-device-only", "-o", out, src] + list(extra_flags), stderr=subprocess.DEVNULL)
        text = open(out).read().split("\n")
    kernels, cur = {}, None
    for line in text:
        m = re.match(r"^(_ZN4msne\w+):", line)
        if m:
            mm = re.match(r"_ZN4msne\d+(k_trace_closest|k_trace_shadow)ILb0ELb(\d)EE", m.group(1))   # <STATS = false, INSTANCED>
            cur = ("%s%s" % (mm.group(1), "_instanced" if mm.group(2) == "1" else "")) if mm else None
            if cur: kernels[cur] = [dict(label="entry", valu=0, x4=0, cvt=0, divf=0, f64=0, ends_loop=False, in_loop=False)]
            continue
        if cur is None: continue
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", line)
        if m:
            kernels[cur].append(dict(label=m.group(1), valu=0, x4=0, cvt=0, divf=0, f64=0, ends_loop=False, in_loop="in Loop" in m.group(2) or "Loop Header" in m.group(2)))
            continue
        ins = line.strip().split(" ")[0].split("\t")[0]
        b = kernels[cur][-1]
        if ins.startswith("v_"): b["valu"] += 1
        if ins.startswith("global_load"): b["x4"] += 1           # (any width: the any-hit kernel fetches the tail of a triangle record with a narrower load)
        if ins.startswith("v_cvt_f32_ubyte"): b["cvt"] += 1
        if ins.startswith("v_div_fixup_f32"): b["divf"] += 1
        if ins.endswith("_f64") or "_f64_" in ins: b["f64"] += 1
        if ins.startswith(("s_cbranch_execnz", "s_cbranch_vccnz", "s_cbranch_scc")) and b["in_loop"]: b["last_branch"] = line.strip().split()[-1]
    res = {}
    for k, blocks in kernels.items():
        r = dict(node=0, tri=0, space=0, refill=0)
        node_i = max(range(len(blocks)), key=lambda i: blocks[i]["cvt"])
        r["node"] = blocks[node_i]["valu"]
        # the triangle region: consecutive in-loop blocks after the node block, starting at the one with the 3 loads, up to (not including) the first out-of-line block (f64 fallback)
        i = node_i + 1
        while i < len(blocks) and blocks[i]["x4"] < 3: i += 1
        while i < len(blocks) and blocks[i]["in_loop"] and blocks[i]["f64"] == 0:
            r["tri"] += blocks[i]["valu"]; i += 1
        space_i = [j for j, b in enumerate(blocks) if j < node_i and b["x4"] >= 5 and b["cvt"] == 0]   # TLAS leaf record (4 loads) + the ray direction
        for j, b in enumerate(blocks):
            if j == node_i or b["divf"] < 3: continue
            if space_i and j > space_i[-1]: r["space"] += b["valu"]
            else: r["refill"] += b["valu"]
        for j in space_i: r["space"] += blocks[j]["valu"]
        res[k] = r
    return res


def print_static(st):
    print("# static VALU instructions of the bodies of the production trace kernels (basic blocks of hipcc -S; see static_split)")
    for kname, r in st.items():
        print("%-28s " % kname + " | ".join("%s %d" % (s, r[s]) for s in SECTIONS))


if __name__ == "__main__":
    st = static_split([x for x in sys.argv[1:] if x.startswith("-D")])
    print_static(st)
    if "--static" in sys.argv:
        sys.exit(0)
    import torch  # noqa
    from moonshine_amd import api, scenes
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    w, h, spp = (int(a[0]), int(a[1]), int(a[2])) if len(a) > 2 else (1920, 1080, 4)
    import json
    for name in ("s1", "sky", "s2"):
        c = api.Context()
        s, l = scenes.s2(c, extent=(w, h)) if name == "s2" else scenes.s1(c, extent=(w, h), env="sky" if name == "sky" else "constant")
        c.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
        c.set_profiling(True, True)
        c.render(s, l, launches=spp, readback=False)
        stt = c.stats(); t = c.traversal_counters(); u = c.traversal_lane_use()
        import glob
        cfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_counters_%s.json" % {"s1": "s1", "sky": "s1_sky", "s2": "s2"}[name])))   # SQ_INSTS_VALU per ray of the same kernels: the newest committed PMC pass
        cfile = cfiles[-1] if cfiles else ""
        measured = json.load(open(cfile))["kernels"] if cfile and os.path.exists(cfile) else {}
        for k, rays, nv, nt in (("closest", stt["closest_rays"], t["closest_node_visits"], t["closest_tri_tests"]), ("shadow", stt["shadow_rays"], t["shadow_node_visits"], t["shadow_tri_tests"])):
            x = u[k]; it = max(x["iterations"], 1); prof = t[k + "_profile"]
            body = st["k_trace_%s%s" % (k, "_instanced" if name == "s2" else "")]
            print("%s k_trace_%s: %d rays | %.2f node visits, %.2f triangle tests, %.2f space changes per ray | %.2f wave iterations per 64 rays" % (name.upper(), k, rays, nv / rays, nt / rays, x["space_body"] / rays, it * 64.0 / rays))
            runs = {"node": x["iter_node"], "tri": x["iter_tri"], "space": x["iter_space"]}
            lanes = {"node": x["node_body"], "tri": x["tri_body"], "space": x["space_body"]}
            tot = measured.get("k_trace_" + k, {}).get("valu_wave_instructions_per_unit")
            acc = 0.0
            for b in ("node", "tri", "space"):
                if not body[b]: continue
                wi = body[b] * runs[b] / rays; acc += wi
                dense = body[b] * (lanes[b] / 64.0) / rays
                print("   %-6s %3d VALU | runs in %5.1f %% of the iterations with %4.1f of 64 lanes | %6.2f wave-instructions per ray%s | at 64 lanes: %5.2f (-%.2f)"
                      % (b, body[b], 100.0 * runs[b] / it, lanes[b] / max(runs[b], 1), wi, (" (%4.1f %% of the measured %.1f)" % (100.0 * wi / tot, tot)) if tot else "", dense, wi - dense))
            if tot:
                print("   rest   (pops, votes, refill %d VALU per new ray, queue, tails): %.2f wave-instructions per ray (%.1f %%)" % (body["refill"], tot - acc, 100.0 * (tot - acc) / tot))
            cyc = {n: prof[n] for n in ("pop", "refill", "node", "tri")}; tc = float(sum(cyc.values())) or 1.0
            print("   wave cycles between the laps of the STATS build: pop %.1f %% | refill + tails %.1f %% | votes + space + node %.1f %% | triangle %.1f %%" % tuple(100.0 * cyc[n] / tc for n in ("pop", "refill", "node", "tri")))
        del c
