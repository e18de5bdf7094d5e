#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's north-star workloads, one process per GPU.

  metric    : Mrays/s (+ ms/frame) — rays = every BVH traversal: primary + bounce + shadow
  --gpus 1  : configs[3] — the ~1 M-triangle Sponza-class atrium (18 textures, Disney materials incl. glass, 2 quad
              lights + env map), 1920x1080, 4 spp, max_depth 5, rr_depth 3
  --gpus N>1: configs[4] — the same scene at 3840x2160, 4 spp, STRONG scaling: the one 4K frame is cut into 32x32 tiles
              dealt to the N ranks by a fixed permutation; every rank accumulates its tiles for the 4 spp and the
              accumulated image is all-gathered over RCCL (one collective per frame, inside libhalart.so) and
              de-interleaved on every rank.  (BENCH_WORKLOAD=configs4 renders that frame on one GPU: the base of the curve;
              the default N = 1 run also reports it under `secondary`.)
  step      : one frame = update_batch(4) == 4 x update() (the reference renders 1 spp per update,
              src/rt_renderer.rs:458-464; the four samples travel through the wavefront kernels together) + render()

Prints ONE JSON line (rank 0).  `roofline` prices the dominant traversal kernel with the algorithmic-bytes figure of
DESIGN.md §4 from counts and launch times measured in this run; `cpu_baseline` times the CPU oracle on the same scene on
the host cores (N = 1 only); `secondary` (N = 1 only, outside the timed region) holds configs[1] and configs[4]-on-one-GPU.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

TILE = 32
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def source_hash():
    """hash of the device code of libhalart.so and of the host code that launches it (every .hip and .h of csrc/, the Makefile, the C
    ABI header; not the file loaders): a committed PMC traffic figure is only quoted for the code it was collected from"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "hala-renderer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")) or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "halart.h"), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(cfg):
    """The oracle (CPU restatement of the same rendering spec, its own binned-SAH BVH) on this box's host cores, on the
    workload's own scene and frame.  Threads = the cores this process may run on, capped at 32 (the box is shared)."""
    import oracle_lib as O
    osc = O.OracleScene(cfg["scene"], envmap=cfg["env"])
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 32))
    W, Hh, spp = cfg["width"], cfg["height"], cfg["spp"]
    osc.render(W, Hh, frames=1, max_depth=cfg["max_depth"], rr_depth=cfg["rr_depth"], rect=(0, 0, 8, 8), threads=threads)  # builds the tree
    frames = 2  # two frames of the workload (8 spp): ~10-25 s of CPU work
    t0 = time.perf_counter()
    _, st = osc.render(W, Hh, frames=spp * frames, max_depth=cfg["max_depth"], rr_depth=cfg["rr_depth"], threads=threads)
    dt = time.perf_counter() - t0
    rays = st.rays_closest + st.rays_shadow
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"oracle (oracle/oracle_render.cpp, OpenMP, {threads} threads) on the same scene, env map and camera: {frames} frames of "
                      f"{W}x{Hh} at {spp} spp = {rays} rays in {dt:.2f} s ({dt * threads:.0f} core-seconds)",
            "ms_per_frame": round(dt * 1e3 / frames, 1)}


class Run:
    """one renderer on one workload; measure() times K frames bracketed by barrier + synchronize on both sides"""

    def __init__(self, H, cfg, local_rank, rank, world, dist, torch, instancing=None):
        self.cfg, self.dist, self.torch, self.world = cfg, dist, torch, world
        self.r = H.HalaRenderer("bench", cfg["width"], cfg["height"], cfg["max_depth"], cfg["rr_depth"], False, False, False, 0,
                                device_ordinal=local_rank)
        r = self.r
        if instancing is not None:  # None: the library's automatic choice (one tree over all triangles, flattened, at this size)
            r.set_build_options(instancing=instancing)
        if world > 1:
            r.set_tile_shard(rank, world, TILE)
        if cfg["env"] is not None:
            r.set_envmap(cfg["env"], 0.0)
        r.set_scene(cfg["scene"])
        t0 = time.perf_counter()
        r.commit()
        self.commit_ms = (time.perf_counter() - t0) * 1e3
        self.gather = None
        if world > 1:
            from hala_renderer_amd.dist import TileGather
            # one all-gather per finished frame: the accumulated colour image; albedo / normal are gathered the
            # same way when save_images needs them (TileGather(aovs=(0, 1, 2)))
            # BENCH_EXCHANGE=torch hands the exchange to torch.distributed (same pipeline); unset: RCCL inside the library, and "torch" only
            # if some rank cannot create the library's communicator (reported in config.exchange)
            self.gather = TileGather(r, local_rank, aovs=(r.ACCUM,), exchange=os.environ.get("BENCH_EXCHANGE") or None)
        self.sync_gather = bool(os.environ.get("BENCH_SYNC_GATHER"))

    def step(self):
        r, g = self.r, self.gather
        r.reset_accumulation()  # a frame restarts the accumulation: frame_index 0..spp-1 (same work every step)
        r.update_batch(self.cfg["spp"])
        if g is not None:
            # pipelined: the collective of frame k runs beside the rendering of frame k + 1 and is waited for by the next
            # begin() / the closing fence (BENCH_SYNC_GATHER=1: wait right here)
            if self.sync_gather:
                g.gather()
            else:
                g.begin()
        r.render()

    def fence(self):
        if self.gather is not None:
            self.gather.finish()  # the last frame's all-gather and de-interleave are part of the timed region
        self.r.wait_idle()
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def measure(self, steps, warmup, timing_period=1, fusion=2):
        """fusion 2: the timed frames run the same fused launches as the untimed ones (traverse_fused_* statistics);
        fusion 0: one launch per pass (traverse_closest_* / traverse_shadow_*: each symbol with the chip to itself)"""
        r = self.r
        r.set_pass_fusion(fusion)
        r.set_launch_timing_period(timing_period)
        for _ in range(warmup):
            self.step()
        self.fence()
        s0 = r.statistics()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.fence()
        dt = time.perf_counter() - t0
        s1 = r.statistics()
        return dt, s0, s1

    def count(self):
        """one extra frame with the counting kernels (identical traversal; tests pin those counts to the oracle's)"""
        r = self.r
        r.set_pass_fusion(0)
        r.set_launch_timing_period(1)
        r.set_counting(True)
        c0 = r.statistics()
        self.step()
        self.fence()
        c1 = r.statistics()
        r.set_counting(False)
        return c0, c1

    def close(self):
        self.r.close()


def kernel_report(timed_stats, unfused_stats, counts, staged, steps):
    """per-kernel-symbol figures.  timed_stats (s0, s1): the timed region — its timed frames ran fused launches (traverse_fused_*);
    unfused_stats (u0, u1): frames with one launch per pass (each symbol with the chip to itself); counts (c0, c1): the counting frame
    (nodes visited / triangles tested per ray kind; same traversal)"""
    s0, s1 = timed_stats
    u0, u1 = unfused_stats
    c0, c1 = counts
    f = lambda name: getattr(s1, name) - getattr(s0, name)  # noqa: E731
    d = lambda name: getattr(u1, name) - getattr(u0, name)  # noqa: E731
    c = lambda name: getattr(c1, name) - getattr(c0, name)  # noqa: E731
    st = "true" if staged else "false"
    out = {}
    # closest-hit traversal runs as two symbols: k_trace_primary (depth 0: camera rays generated in place) and k_trace_batch (bounce queues)
    n_b = max(c("rays_closest_counted") - c("rays_primary_counted"), 1)
    nodes_b = (c("nodes_closest_total") - c("nodes_primary_total")) / n_b
    tris_b = (c("tris_closest_total") - c("tris_primary_total")) / n_b
    n_p = max(c("rays_primary_counted"), 1)
    nodes_p, tris_p = c("nodes_primary_total") / n_p, c("tris_primary_total") / n_p
    n_s = max(c("rays_shadow_counted"), 1)
    nodes_s, tris_s = c("nodes_shadow_total") / n_s, c("tris_shadow_total") / n_s
    ms_b = d("traverse_closest_ms_total") - d("traverse_primary_ms_total")
    l_b = d("traverse_closest_launches") - d("traverse_primary_launches")
    rays_b = d("rays_closest_timed") - d("rays_primary_timed")
    bpr = lambda fixed, nodes, tris: fixed + 64.0 * nodes + 48.0 * tris  # noqa: E731  (algorithmic bytes per ray)

    def entry(symbol, what, fixed_bytes, nodes, tris, ms, launches, rays):
        b = bpr(fixed_bytes, nodes, tris)
        avg = ms / max(launches, 1)
        rpl = rays / max(launches, 1)
        ach = b * rpl / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
        return {"kernel": symbol, "what": what, "bytes_per_ray": round(b, 1), "nodes_per_ray": round(nodes, 3), "tris_per_ray": round(tris, 3),
                "avg_launch_ms": round(avg, 5), "launches": int(launches), "rays_per_launch": round(rpl, 1),
                "grays_per_s_in_kernel": round(rays / max(ms, 1e-9) / 1e6, 3), "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4)}

    out["batch"] = entry(f"rt::k_trace_batch<false, false, {st}, false>", "closest-hit traversal of the bounce-ray queues (depth >= 1): 32 B ray read + 16 B hit write per ray",
                         48.0, nodes_b, tris_b, ms_b, l_b, rays_b)
    out["primary"] = entry(f"rt::k_trace_primary<false, {st}>", "depth 0: camera rays generated in the lanes that trace them, 16 B hit write per ray",
                           16.0, nodes_p, tris_p, f("traverse_primary_ms_total") + d("traverse_primary_ms_total"),
                           f("traverse_primary_launches") + d("traverse_primary_launches"), f("rays_primary_timed") + d("rays_primary_timed"))
    out["shadow"] = entry(f"rt::k_trace_shadow<false, {st}, false>", "any-hit traversal of the NEE connections: 48 B entry + 12 B radiance read per connection (+ 12 B add when unoccluded)",
                          60.0, nodes_s, tris_s, d("traverse_shadow_ms_total"), d("traverse_shadow_launches"), d("rays_shadow_timed"))
    # the fused launch: both shadow passes of bounce d + the closest-hit pass of bounce d + 1 — the symbol that dominates a frame
    rfc, rfs = f("rays_fused_closest_timed"), f("rays_fused_shadow_timed")
    ms_f, l_f = f("traverse_fused_ms_total"), f("traverse_fused_launches")
    bytes_f = rfc * bpr(48.0, nodes_b, tris_b) + rfs * bpr(60.0, nodes_s, tris_s)
    ach_f = bytes_f / (ms_f * 1e-3) / 1e9 if ms_f > 0 else 0.0
    out["fused"] = {"kernel": f"rt::k_trace_shadow_then_batch<{st}, false>",
                    "what": "ONE persistent launch per bounce: the light and environment connections of bounce d (any hit) and the bounce rays of bounce d + 1 (closest hit)",
                    "bytes_per_ray": {"bounce_ray": round(bpr(48.0, nodes_b, tris_b), 1), "connection": round(bpr(60.0, nodes_s, tris_s), 1)},
                    "rays_per_launch": {"bounce_rays": round(rfc / max(l_f, 1), 1), "connections": round(rfs / max(l_f, 1), 1)},
                    "algorithmic_bytes_per_launch": round(bytes_f / max(l_f, 1), 1),
                    "avg_launch_ms": round(ms_f / max(l_f, 1), 5), "launches": int(l_f),
                    "grays_per_s_in_kernel": round((rfc + rfs) / max(ms_f, 1e-9) / 1e6, 3), "achieved": round(ach_f, 1), "frac": round(ach_f / HBM_PEAK_GBS, 4)}
    simt = {}
    for kind in ("closest", "shadow"):
        ws = max(c(f"wave_steps_{kind}_total"), 1)
        lp = max(c(f"leaf_passes_{kind}_total"), 1)
        simt[kind] = {"node_path_lanes_of_64": round(c(f"nodes_{kind}_total") / ws, 1), "leaf_passes_per_wave_step": round(c(f"leaf_passes_{kind}_total") / ws, 2),
                      "leaf_path_lanes_of_64": round(c(f"leaf_lanes_{kind}_total") / lp, 1)}
    out["simt"] = simt
    sh_ms, sh_l = f("shade_ms_total") + d("shade_ms_total"), f("shade_launches") + d("shade_launches")
    out["shade"] = {"kernel": "rt::k_shade<PRIMARY>", "avg_launch_ms": round(sh_ms / max(sh_l, 1), 5), "launches": int(sh_l)}
    uf = max(d("traverse_primary_launches"), 1)  # frames with one launch per pass
    ff = max(f("traverse_primary_launches"), 1)  # timed frames of the timed region (fused launches)
    out["ms_per_frame_by_kernel"] = {
        "one_launch_per_pass": {"closest": round(d("traverse_closest_ms_total") / uf, 4), "shade": round(d("shade_ms_total") / uf, 4),
                                "shadow": round(d("traverse_shadow_ms_total") / uf, 4), "frames": int(uf)},
        "fused_launches": {"primary": round(f("traverse_primary_ms_total") / ff, 4), "shade": round(f("shade_ms_total") / ff, 4),
                           "fused_traversal": round(ms_f / ff, 4), "last_bounce_shadow": round(f("traverse_shadow_ms_total") / ff, 4), "frames": int(ff)},
        "gpu_total": round(f("gpu_ms_total") / max(steps, 1), 4)}  # frame begin -> frame end events, all frames of the timed region
    return out


def valu_note(counters):
    """what binds the traversal kernels: VALU issue slots x active lanes (PMC, committed with the source hash it was collected from)"""
    if not counters or counters.get("valu_issue_utilisation") is None or counters.get("active_lanes_per_valu_instruction") is None:
        return None
    util, lanes = counters["valu_issue_utilisation"], counters["active_lanes_per_valu_instruction"]
    return {"issue_utilisation": util, "active_lanes_of_64": lanes, "lane_issue_frac": round(util * lanes / 64.0, 4),
            "note": "issue_utilisation = SQ_INSTS_VALU / (1024 SIMDs x kernel cycles / 4) with kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs (a SIMD issues one "
                    "wave64 VALU instruction per 4 cycles); active_lanes = SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU; their product / 64 = the share of the chip's "
                    "vector lane-issue capacity doing work — the limit that binds these kernels (the tree is cache-resident)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python3 bench.py --gpus N` from a bare shell: start the N ranks ourselves, exactly as the driver would (one process per GPU
        # under torch.distributed.run, rendezvous on 127.0.0.1).  This parent never touches the GPU (no torch import, no HIP call): it
        # only waits for the launcher — a child process, not an exec — and passes rank 0's JSON line and the exit code on.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")

    import torch

    import hala_renderer_amd as H
    from hala_renderer_amd import workloads

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libhalart has no CPU path")
    # rehearsal knobs for a 1-GPU box (not used by the driver): BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0 and
    # BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device) so that the N>1 code path can be exercised
    if os.environ.get("BENCH_SINGLE_DEVICE"):
        local_rank = 0
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    index = 4 if (world > 1 or os.environ.get("BENCH_WORKLOAD") == "configs4") else 3
    cfg = workloads.baseline_config(index)
    # BENCH_INSTANCING=1 (a probe, not used by the driver): the main workload on a two-level tree, with the per-kernel report of that form
    run = Run(H, cfg, local_rank, rank, world, dist, torch, instancing=True if os.environ.get("BENCH_INSTANCING") else None)
    # per-launch HIP events (the roofline's avg_launch_ms) on two frames of the timed region (still live, still inside it): a timed frame
    # issues one launch per pass, the others fuse the shadow passes of a bounce with the next bounce's closest-hit pass (renderer.hip)
    # (two timed frames from 8 steps on, else exactly one: any `p` consecutive updates hold one multiple of `p`)
    period = int(os.environ.get("BENCH_TIMING_PERIOD", str(args.steps // 2 if args.steps >= 8 else max(1, args.steps))))
    dt, s0, s1 = run.measure(args.steps, args.warmup, timing_period=period, fusion=2)
    rays_local = s1.rays_total - s0.rays_total
    t_local = torch.tensor([dt, float(rays_local)], dtype=torch.float64, device=f"cuda:{local_rank}")
    if dist is not None:
        tmax = t_local.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        rsum = t_local.clone()
        dist.all_reduce(rsum, op=dist.ReduceOp.SUM)
        dt_all, rays_all = float(tmax[0]), float(rsum[1])
    else:
        dt_all, rays_all = dt, float(rays_local)

    # ---- roofline of the dominant kernel (rank 0's launches), from this run's own counts and launch times --------------
    # outside the timed region: two frames with one launch per pass (every traversal symbol timed with the chip to itself), then one
    # frame with the counting kernels
    _, u0, u1 = run.measure(2, 0, timing_period=1, fusion=0)
    counts = run.count()
    info = run.r.bvh_info()
    tree_bytes_flat = info.tree_bytes
    staged = info.lds_node_count > 0
    kr = kernel_report((s0, s1), (u0, u1), counts, staged, args.steps)
    # a second commit() of the same scene, outside the timed region: the steady-state build (the first one also pays the arena allocations and the upload)
    rebuild_ms = None
    if world == 1:
        run.r.wait_idle()
        t0 = time.perf_counter()
        run.r.commit()
        run.r.wait_idle()
        rebuild_ms = (time.perf_counter() - t0) * 1e3

    out = None
    if rank == 0:
        traffic = None
        tfile = "profiles/r03_traffic_config3.json"
        tnote = f"not collected in this run (PMC counters need rocprofv3: scripts/profile_round.sh writes {tfile})"
        tpath = os.path.join(ROOT, tfile)
        if index == 3 and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("source_hash") == source_hash():
                    traffic = tj
                    tnote = f"collected by scripts/profile_round.sh from this exact source (hash {tj.get('source_hash')}, commit {tj.get('commit')}), separate --pmc passes, guide's gfx950 correction"
                else:
                    tnote = f"{tfile} was collected from other sources (hash {tj.get('source_hash')} != {source_hash()}): not quoted"
            except Exception as e:  # noqa: BLE001
                tnote = f"unreadable: {e}"
        # the symbol that dominates a frame: the fused launch (both shadow passes of bounce d + the closest-hit pass of bounce d + 1),
        # timed by HIP events on the renderer's stream inside the timed region; the unfused symbols are listed beside it
        b = kr["fused"] if kr["fused"]["launches"] else kr["batch"]
        tb = traffic["hbm_bytes_per_launch"] if traffic and traffic.get("kernel", "").startswith(b["kernel"].split("<")[0]) else None
        counters = ({k: traffic.get(k) for k in ("l2_hit_rate", "valu_issue_utilisation", "active_lanes_per_valu_instruction", "wait_any_share_of_wave_cycles",
                                                "FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch")} if traffic else None)
        roof = {"bound": "hbm", "kernel": b["kernel"] + " — " + b["what"],
                "achieved": b["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b["frac"],
                "traffic": tb, "traffic_note": tnote,
                "traffic_GBps": (round(tb / (b["avg_launch_ms"] * 1e-3) / 1e9, 1) if tb and b["avg_launch_ms"] > 0 else None),
                "traffic_frac_of_peak": (round(tb / (b["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tb and b["avg_launch_ms"] > 0 else None),
                "counters": counters,
                "valu": valu_note(counters),
                "note": "achieved = ALGORITHMIC bytes (SURVEY 8d formula with this library's formats: 64 B per 4-wide node visited, 48 B per triangle tested, "
                        "+ the ray / hit / connection record bytes; per ray kind) x the rays the launches traced from each queue / the time of those launches: "
                        "HIP events on the renderer's stream around every launch of every `timing_period`-th frame of the timed region (`launches` = the "
                        "launches so measured; profiles/r03_*_kernel_stats_bench.csv holds the rocprofv3 average of the same command).  Node and "
                        "triangle counts per ray kind come from one extra frame with the counting kernels in this run.  The tree "
                        f"({info.node_count * 64 / 1e6:.0f} MB of nodes + {info.triangle_count * 48 / 1e6:.0f} MB of triangles) sits in L2 / Infinity Cache, so `traffic` (fabric bytes by PMC) is far below the algorithmic bytes: "
                        "frac prices useful work against the HBM peak — a work rate, not a distance to a hardware limit; traffic_frac_of_peak is what the memory side really carries; "
                        "`valu` is the limit that binds (vector issue slots x active lanes); `simt`: active lanes per wave on the two code paths.",
                "bytes_per_ray": b["bytes_per_ray"], "avg_launch_ms": b["avg_launch_ms"], "launches": b["launches"], "timing_period": period,
                "rays_per_launch": b["rays_per_launch"], "grays_per_s_in_kernel": b["grays_per_s_in_kernel"], "simt": kr["simt"],
                "unfused_kernels": {"note": "two extra frames with one launch per pass (hala_rt_set_pass_fusion 0), outside the timed region: every symbol timed with the chip to itself",
                                    "batch": kr["batch"], "shadow": kr["shadow"]},
                "primary_kernel": kr["primary"], "shade_kernel": kr["shade"],
                "ms_per_frame_by_kernel": kr["ms_per_frame_by_kernel"]}
        out = {
            "metric": "Mrays/s", "value": round(rays_all / dt_all / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_all / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["name"], "resolution": [cfg["width"], cfg["height"]], "spp": cfg["spp"], "max_depth": cfg["max_depth"], "rr_depth": cfg["rr_depth"],
                       "triangles": int(info.triangle_count), "bvh_nodes": int(info.node_count), "bvh_build_ms": round(run.commit_ms, 2), **({"bvh_rebuild_ms": round(rebuild_ms, 2)} if rebuild_ms is not None else {}),
                       "parallelism": f"pixel-tile shard {TILE}x{TILE} of the one frame over {world} rank(s)" + (", one RCCL all-gather of the accumulated image per frame" if world > 1 else ""),
                       **({"exchange": run.gather.exchange + (f" (fallback: {run.gather.fallback_reason})" if run.gather.fallback_reason else "")} if run.gather is not None else {}),
                       "rays_per_frame": int(rays_all / args.steps), "ms_per_frame": round(dt_all / args.steps * 1e3, 4),
                       "scaling_note": "N = 1 renders configs[3] (1920x1080); N > 1 renders configs[4] (3840x2160) strong-scaled; the single-GPU time of the 4K frame is under secondary.configs4_on_1_gpu"},
            "roofline": roof,
        }
    run.close()

    if rank == 0 and world == 1:
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        else:
            out["cpu_baseline"] = None
        secondary = {}
        if not args.no_secondary and index == 3:
            # configs[4]'s frame on this one GPU: the base of the strong-scaling curve the driver's N > 1 runs continue
            c4 = workloads.baseline_config(4)
            c4["scene"], c4["env"] = cfg["scene"], cfg["env"]  # same scene objects (16:9 either way)
            r4 = Run(H, c4, local_rank, 0, 1, None, torch)
            d4, a0, a1 = r4.measure(3, 1, timing_period=0)
            secondary["configs4_on_1_gpu"] = {"workload": c4["name"] + " on ONE GPU (no shard, no collective)", "steps": 3,
                                              "value": round((a1.rays_total - a0.rays_total) / d4 / 1e6, 2), "unit": "Mrays/s", "ms_per_frame": round(d4 / 3 * 1e3, 3)}
            r4.close()
            # configs[3] as a two-level tree (hala_rt_build_options::instancing = 2: the reference's BLAS / TLAS split, RENDER_SPEC 4.5)
            info_flat = None
            try:
                rt2 = Run(H, cfg, local_rank, 0, 1, None, torch, instancing=True)
                d2, t0_, t1_ = rt2.measure(6, 2, timing_period=0)
                i2 = rt2.r.bvh_info()
                secondary["configs3_two_level"] = {"workload": cfg["name"] + " — two-level tree: every instanced primitive stored once, instance levels on top",
                                                   "steps": 6, "value": round((t1_.rays_total - t0_.rays_total) / d2 / 1e6, 2), "unit": "Mrays/s", "ms_per_frame": round(d2 / 6 * 1e3, 3),
                                                   "stored_triangles": int(i2.stored_triangle_count), "instance_refs": int(i2.instance_ref_count),
                                                   "tree_bytes": int(i2.tree_bytes), "tree_bytes_flattened": int(tree_bytes_flat),
                                                   "note": "images of the two forms agree to rounding, each is held bit for bit to the oracle following the same rule (tests/test_gpu_parity.py)"}
                rt2.close()
            except Exception as e:  # noqa: BLE001
                secondary["configs3_two_level"] = {"error": str(e)}
            # configs[1]: the 32-triangle Cornell box whose BVH lives in LDS (round 1's headline)
            c1 = workloads.baseline_config(1)
            r1 = Run(H, c1, local_rank, 0, 1, None, torch)
            d1, b0, b1 = r1.measure(20, 3, timing_period=4, fusion=1)  # its timed frames: one launch per pass
            k1 = kernel_report((b0, b0), (b0, b1), r1.count(), True, 20)
            secondary["configs1"] = {"workload": c1["name"], "steps": 20, "value": round((b1.rays_total - b0.rays_total) / d1 / 1e6, 2), "unit": "Mrays/s",
                                     "ms_per_frame": round(d1 / 20 * 1e3, 4),
                                     "batch_kernel": k1["batch"], "shadow_kernel": k1["shadow"], "shade_kernel": k1["shade"], "simt": k1["simt"],
                                     "note": "the whole BVH (1.9 KB) is staged in LDS: the algorithmic-bytes fraction is a rate of useful work, not an HBM bandwidth, and can exceed 1"}
            r1.close()
        out["secondary"] = secondary
    elif rank == 0:
        out["cpu_baseline"] = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
