#!/usr/bin/env python3
"""Benchmark of the ray-triangle hot path on MI355X.

One "step" = one full frame of the pixel loop (driver.c:126-193) over a synthetic scene that
is already resident in HBM: per-frame vertex transform + sweep-buffer rebuild, primary
closest-hit sweep, fp64 confirmation, shading epilogue with its shadow-ray sweeps, and -- for
N > 1 -- the gather of the rows to rank 0 over RCCL.  The image is sharded by interleaved
16-row blocks (strong scaling: the frame is fixed, each GPU holds the full mesh); a frame
sequence (teapot_seq_4k) is sharded by frames instead (weak scaling, no exchange).

Prints ONE JSON line on rank 0.  `value` is whole-job Mrays/s (primary + shadow rays of the
frame divided by the frame time).  Along ride: pairs decided per second and the roofline of the
sweeps (timed with HIP events on their launch stream inside the library; a "test" of the
pixel-grid and cell-lattice sweeps is a (ray, triangle) pair DECIDED -- most of them by a
separable box test shared by the rows and columns of a block of pixels or direction cells -- not a
pair evaluated on its own; `roofline.bound` and `roofline.hot_loop_only_ms` say how much of the
primary launch that test is), the general
Moller-Trumbore sweep on the same rays (`roofline.general_mt`: every pair evaluated on its own
in fp32 FMAs), and a CPU-oracle baseline timed on this host.

--obj PATH renders a user-supplied Wavefront file (loaded by the library's own loader) instead
of the workload's procedural mesh; a path that does not exist prints "asset not available" and
keeps the stand-in.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TAN_HALF = math.tan(30 * math.pi / 180)          # driver.c:22,25
PEAK_FP32_TFLOPS = 157.3                         # MI355X vector fp32 (256 CU x 256 flop/clk x 2.4 GHz)
PEAK_VALU_TINSTR = PEAK_FP32_TFLOPS / 2          # VALU lane-instructions/s: the same issue slots, one FMA = 2 flop

WORKLOADS = {
    # name: (mesh kind, triangles, W, H)
    "soup1m_4k": ("soup", 1_000_000, 3840, 2160),
    "mesh100k_4k": ("torus", 100_000, 3840, 2160),
    "bunny_class_4k": ("sphere", 69_451, 3840, 2160),
    "teapot_class_480p": ("torus", 6_320, 640, 480),
    "soup100k_1080p": ("soup", 100_000, 1920, 1080),
    # config 5: 360-frame orbit, one M3d transform per frame (step k renders frame k)
    "teapot_seq_4k": ("torus", 6_320, 3840, 2160),
}


def make_mesh(kind: str, tris: int):
    from raytracer_amd import scenes
    if kind == "soup":
        return scenes.triangle_soup(tris, seed=12345)
    if kind == "torus":
        return scenes.torus(tris)
    return scenes.sphere(tris)


def load_mesh(workload: str, obj_path: str | None):
    """The mesh of a run: the user's .obj through the library's loader (rt_obj_load) when the
    file exists, else the workload's procedural stand-in.  Returns (mesh for host-side use,
    ObjFile or None, note)."""
    kind, tris, _, _ = WORKLOADS[workload]
    if obj_path:
        if os.path.exists(obj_path):
            from raytracer_amd import kernel
            obj = kernel.ObjFile(obj_path)
            return obj.mesh(), obj, f"{os.path.basename(obj_path)}: {obj.num_tris} triangles (rt_obj_load)"
        print(f"bench.py: asset not available: {obj_path}; using the procedural stand-in "
              f"({kind}, {tris} triangles)", file=sys.stderr)
        return make_mesh(kind, tris), None, f"asset not available ({os.path.basename(obj_path)}): procedural stand-in"
    return make_mesh(kind, tris), None, "procedural"


def general_mt(sc, W: int, H: int, dev, hit_primary=None) -> dict:
    """The frame's primary rays given as rays in general position (S, E) to rt_trace_rays: the
    sweep that evaluates every (ray, triangle) pair on its own with the Moller-Trumbore
    determinants in fp32 FMAs (rt_sweep.hip, moment form: 21 executed flop per pair), timed by
    the library's HIP events around the launch."""
    import ctypes as C
    import torch
    from raytracer_amd import _lib, kernel
    ez = (W // 2) / TAN_HALF                                    # driver.c:133
    xs = (torch.arange(W, device=dev, dtype=torch.float64) - W // 2).expand(H, W)
    ys = (torch.arange(H, device=dev, dtype=torch.float64) - H // 2)[:, None].expand(H, W)
    E = torch.stack([xs, ys, torch.full((H, W), ez, device=dev, dtype=torch.float64)], dim=-1).reshape(-1, 3).contiguous()
    S = torch.zeros_like(E)
    sc.trace_rays(S[:4096], E[:4096])                           # the kernel's first launch
    kernel.profile_enable(True)
    kernel.profile_reset()
    hit, _, _ = sc.trace_rays(S, E)
    prof = kernel.profile_get()
    kernel.profile_enable(False)
    form, tile, rpl, flop = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    _lib.load().rt_sweep_config(C.byref(form), C.byref(tile), C.byref(rpl), C.byref(flop))
    ms = prof.sweep_ms
    rate = prof.sweep_tests / (ms * 1e-3) if ms > 0 else 0.0
    achieved = rate * flop.value / 1e12
    out = {"kernel": "k_sweep<closest hit> (rays in general position, every pair evaluated on its own)",
           "rays": int(S.shape[0]), "ms": ms, "tests": float(prof.sweep_tests), "tests_per_s": rate,
           "flop_per_test": flop.value, "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
           "frac": achieved / PEAK_FP32_TFLOPS, "target_50pct_fp32_flops_met": bool(achieved / PEAK_FP32_TFLOPS >= 0.5),
           "rays_per_lane": rpl.value, "tile_tris": tile.value}
    if hit_primary is not None:
        out["hit_equals_pixel_grid_sweep"] = bool(torch.equal(hit.reshape(H, W), hit_primary))
    return out


def measured_traffic(workload: str, cfg: str) -> dict | None:
    """HBM-side bytes of one closest-hit sweep launch of this workload and kernel configuration,
    from the committed rocprofv3 counter passes (profiles/traffic.json; FETCH_SIZE and WRITE_SIZE
    are collected in runs of their own, so they cannot be read live).  None when no pass was
    recorded for this exact configuration."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            table = json.load(f)
    except OSError:
        return None
    return table.get(f"{workload}/{cfg}")


def counted_valu(workload: str, kernel_prefix: str) -> dict | None:
    """Vector instructions one launch of the primary sweep executes, whole launch (rare path
    included), as counted by rocprofv3 (SQ_INSTS_VALU of the committed counter pass: counters are
    collected in runs of their own and cannot be read live).  None when there is no pass for this
    workload and kernel."""
    if workload != "soup1m_4k" or not kernel_prefix.startswith("k_sweep_raster_wave"):
        return None
    path = os.path.join(ROOT, "profiles", "r03_bench_soup1m_4k", "raster_counters.txt")
    try:
        vals = [float(m) for m in re.findall(r"SQ_INSTS_VALU=([0-9.e+]+)", open(path).read())]
    except OSError:
        return None
    if not vals:
        return None
    return {"wave_instructions": max(vals), "source": "profiles/r03_bench_soup1m_4k/raster_counters.txt"}


def cpu_baseline(mesh, frame, W, H, budget_s: float = 25.0):
    """CPU oracle (strict fp64 restatement of the reference's formulas, OpenMP over rows) on
    a bounded row sample of the same frame; primary rays only."""
    from oracle import oracle as O
    from raytracer_amd import kernel, parallel
    X, Y, Z = O.transform(mesh, kernel.frame_matrix(frame, "obmat"))
    osc = O.Scene(mesh, X, Y, Z)
    ofr = O.frame_from(frame)
    import ctypes as C
    threads = min(O.lib().orc_num_threads(), int(os.environ.get("OMP_NUM_THREADS", "16")), 64)
    # calibrate on a strip of a narrower image (same per-ray cost), then size the sample
    cal_w = 16 * threads
    y_mid = H // 2
    t0 = time.time()
    O.lib().orc_render_rows(C.byref(osc.s), C.byref(ofr), cal_w, H, TAN_HALF, y_mid, y_mid + 1, 1, 0,
                            threads, None, None, None, None)
    per_ray = max(time.time() - t0, 1e-4) / cal_w          # wall seconds per ray, all threads busy
    rows = max(min(int(budget_s / (per_ray * W)), H), 1)
    step = max(H // rows, 1)
    ys = list(range(step // 2, H, step))[:rows]
    t0 = time.time()
    O.lib().orc_render_rows(C.byref(osc.s), C.byref(ofr), W, H, TAN_HALF, ys[0], ys[-1] + 1, step, 0,
                            threads, None, None, None, None)
    dt = time.time() - t0
    rays = len(ys) * W
    # the reference itself is single-threaded: one thread on a short strip of the same row
    one_w = 64
    t0 = time.time()
    O.lib().orc_render_rows(C.byref(osc.s), C.byref(ofr), one_w, H, TAN_HALF, y_mid, y_mid + 1, 1, 0,
                            1, None, None, None, None)
    dt1 = max(time.time() - t0, 1e-6)
    return {
        "value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
        "tests_per_s": rays * mesh.num_tris / dt,
        "single_thread_tests_per_s": one_w * mesh.num_tris / dt1,
        "sample": f"{len(ys)} of {H} rows (every {step}th) x {W} primary rays x {mesh.num_tris} triangles, "
                  f"fp64, {dt:.1f} s",
    }


def mix_alone(name: str, tests_per_s: float) -> dict:
    """Speed of light of a sweep's instruction mix (`raster64`, `apex8`, ...), from the committed
    micro-benchmark output: the hot loop's instruction sequence with nothing around it."""
    path = os.path.join(ROOT, "profiles", "valu_rate_gfx950.txt")
    if not os.path.exists(path):
        return {}
    for line in open(path):
        if line.split()[:1] == [name]:
            rate = float(line.split()[-2])
            return {"mix_alone_tests_per_s": rate, "frac_of_mix_alone": tests_per_s / rate,
                    "mix_alone_source": "profiles/valu_rate_gfx950.txt"}
    return {}


ROCPROF_DIR = "profiles/r03_bench_soup1m_4k"      # tools/profile_bench.sh over this program's default run


def rocprof_avg_ms(workload: str, kernel_prefix: str, minus_fastest: bool = False) -> dict | None:
    """Average duration of a kernel in the committed rocprofv3 --kernel-trace --stats summary of
    this program's default run, for the cross-check of the HIP-event figure (None for other
    workloads, or when the summary does not list the kernel).  minus_fastest: the run holds one
    launch of the kernel that does less than the others (the primary sweep with its rare path
    off, for hot_loop_only_ms); the summary's total and minimum give the average without it."""
    if workload != "soup1m_4k":
        return None
    path = os.path.join(ROOT, ROCPROF_DIR, "kernel_stats.csv")
    try:
        import csv
        for row in csv.DictReader(open(path)):
            if kernel_prefix in row["Name"]:
                calls = int(row["Calls"])
                if minus_fastest and calls > 1:
                    return {"avg_ms": (float(row["TotalDurationNs"]) - float(row["MinNs"])) / (calls - 1) / 1e6,
                            "calls": calls - 1, "left_out": "the fastest launch (rare path off)",
                            "hot_loop_only_ms": float(row["MinNs"]) / 1e6, "source": f"{ROCPROF_DIR}/kernel_stats.csv"}
                return {"avg_ms": float(row["AverageNs"]) / 1e6, "calls": calls,
                        "source": f"{ROCPROF_DIR}/kernel_stats.csv"}
    except (OSError, KeyError, ValueError):
        return None
    return None


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="soup1m_4k", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--obj", default=None, metavar="PATH",
                    help="a Wavefront .obj to render instead of the workload's procedural mesh (its .mtl and "
                         "image maps are looked up beside it); a missing file keeps the stand-in")
    ap.add_argument("--no-general-mt", action="store_true",
                    help="skip the general Moller-Trumbore sweep of the frame's primary rays (roofline.general_mt)")
    ap.add_argument("--cpu-budget", type=float, default=25.0)
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group and run the gather even with one rank")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: functional rehearsal of the multi-rank path with several ranks on one "
                         "GPU (the gather goes through host copies); not a measurement")
    ap.add_argument("--single-process", action="store_true",
                    help="without a launcher: this one process drives --gpus devices through "
                         "rt_render_frame_multi (one host thread per device inside the library, rows to "
                         "device 0 over RCCL send/recv) instead of one process per GPU")
    ap.add_argument("--devices", default=None,
                    help="with --single-process: comma-separated HIP ordinals (default 0..gpus-1; a device "
                         "may be named twice for a rehearsal on one GPU)")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed steps rank 0 renders the whole frame alone and compares it "
                         "with the assembled screen")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracer_amd import kernel, parallel

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    single = args.single_process
    devices = ([int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))) if single else None
    if single and (world != 1 or args.workload.endswith("_seq_4k")):
        print("bench.py: --single-process runs without a launcher and renders single frames", file=sys.stderr)
        sys.exit(2)
    if not single and world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; this benchmark has no CPU path", file=sys.stderr)
        sys.exit(2)
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    kind, tris, W, H = WORKLOADS[args.workload]
    mesh, obj, mesh_note = load_mesh(args.workload, args.obj)
    sequence = args.workload.endswith("_seq_4k")
    frames = [kernel.driver_frame(k, 360) for k in range(360)] if sequence else [kernel.driver_frame(0)]
    obmats = [kernel.frame_matrix(f, "obmat") for f in frames]
    frame = frames[0]
    step_no = [0]
    profiled = [False]        # the profiled frame of a --single-process run goes through the plain scene
    gather_ms = [None]        # exchange + placement of the profiled frame (host clock around it, devices idle before)
    sc = kernel.Scene(obj if obj is not None else mesh, device=local_rank)
    ms = kernel.MultiScene(obj if obj is not None else mesh, devices) if single else None
    # A frame sequence is split by frames: rank r renders frames r, r + world, ... whole, and
    # nothing is exchanged (driver.c:196-199 writes one picture per frame; so would every rank).
    frame_parallel = sequence and world > 1
    # Otherwise rows are dealt to the ranks in interleaved blocks of 16 (one rank: the whole image)
    my_rows = parallel.part_rows(H, 1, 0) if frame_parallel else parallel.part_rows(H, world, rank)
    rows = kernel.rows_count(my_rows)
    max_rows = rows if frame_parallel else parallel.max_part_rows(H, world)
    # outputs live in HBM for the whole run; the gather buffer is padded to the largest part
    out = {"hit": torch.empty((rows, W), dtype=torch.int32, device=dev),
           "rgb8": torch.zeros((max_rows, W, 3), dtype=torch.uint8, device=dev)}
    gathered = ([torch.empty((max_rows, W, 3), dtype=torch.uint8, device=dev) for _ in range(world)]
                if (use_dist and rank == 0) else None)
    screen = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev) if (rank == 0 or frame_parallel) else None

    def step():
        k = (step_no[0] * world + rank if frame_parallel else step_no[0]) % len(frames)
        step_no[0] += 1
        if ms is not None and not profiled[0]:
            ms.transform(obmats[k])
            ms.render_frame(frames[k], W, H, TAN_HALF, screen=screen)
            return
        sc.transform(obmats[k])                   # driver.c:110 on the device-resident mesh
        sc.render_rows(frames[k], W, H, TAN_HALF, want=("hit", "rgb8"), rows_set=my_rows,
                       out={"hit": out["hit"], "rgb8": out["rgb8"][:rows]})
        timing_gather = profiled[0] and use_dist and not frame_parallel
        if timing_gather:
            torch.cuda.synchronize()
            dist.barrier()
            t_g = time.perf_counter()
        if frame_parallel:
            sc.place_rows(out["rgb8"], W, H, screen=screen, rows_set=my_rows)
        elif use_dist:
            if args.backend == "gloo":
                host = [torch.empty(out["rgb8"].shape, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
                dist.gather(out["rgb8"].cpu(), host, dst=0)
                if rank == 0:
                    for r in range(world):
                        gathered[r].copy_(host[r])
            elif world > 1:
                parallel.gather_bands(out["rgb8"], H, world, rank, dst=0, out=gathered)   # RCCL over xGMI
            else:
                dist.gather(out["rgb8"], gathered, dst=0)
            if rank == 0:
                for r in range(world):
                    sc.place_rows(gathered[r], W, H, screen=screen, rows_set=parallel.part_rows(H, world, r))
        else:
            sc.place_rows(out["rgb8"], W, H, screen=screen, rows_set=my_rows)
        if timing_gather:
            torch.cuda.synchronize()
            gather_ms[0] = (time.perf_counter() - t_g) * 1e3

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    # One more frame, untimed, with the library's profiling on: HIP events around every sweep
    # launch on its stream and the ray / pair / candidate counts (the events and counter reads
    # cost a sync per launch, which is why the timed steps run without them).
    kernel.profile_enable(True)
    kernel.profile_reset()
    step_no[0] -= 1 if not sequence else 0        # the same frame again, except in a sequence
    profiled[0] = True
    if ms is not None:
        single_screen = screen.clone()
    step()
    fence()
    prof = kernel.profile_get()
    kernel.profile_enable(False)
    # the shape and instruction mix of that frame's sweeps (the library reports its most recent launch)
    import ctypes as C
    from raytracer_amd import _lib
    tile, ppl, spl, chunks = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    f_primary, f_shadow = C.c_double(), C.c_double()
    _lib.load().rt_apex_sweep_config(C.byref(tile), C.byref(ppl), C.byref(f_primary), C.byref(spl),
                                     C.byref(f_shadow), C.byref(chunks))
    mix = [C.c_int() for _ in range(5)]
    _lib.load().rt_raster_sweep_mix(*[C.byref(m) for m in mix])
    regular = C.c_int()
    _lib.load().rt_raster_sweep_form(C.byref(regular))
    # The primary launch once more with its rare path left out (RT_AMD_RASTER_SKIP_RARE, a measurement
    # switch the library reads at every call; the hits of that launch are wrong and are thrown away):
    # what the hot loop alone takes, i.e. how much of the launch is the instruction-bound part.
    hot_only_ms = None
    if world == 1 and not single and not sequence:
        os.environ["RT_AMD_RASTER_SKIP_RARE"] = "1"
        try:
            kernel.profile_enable(True)
            kernel.profile_reset()
            sc.trace_primary(W, H, TAN_HALF)
            p2 = kernel.profile_get()
            hot_only_ms = p2.sweep_ms / max(p2.sweep_launches, 1)
        finally:
            kernel.profile_enable(False)
            del os.environ["RT_AMD_RASTER_SKIP_RARE"]

    # The same steps once more with the library's tile culling on (off by default: the headline
    # is the all-pairs sweep).  Every wave still visits every triangle, but tests a tile's
    # triangles against its whole bundle of rays first; the pictures must be identical.
    dt_culled, culled_same = 0.0, None
    culling_was_on = kernel.tile_culling()
    if not culling_was_on:
        plain_screen = screen.clone() if (screen is not None and not sequence) else None
        kernel.set_tile_culling(True)
        profiled[0] = False
        step_no[0] -= 1 if not sequence else 0
        step()                                    # warm-up: the culled kernels' first launch
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_no[0] -= 1 if not sequence else 0
            step()
        fence()
        dt_culled = time.perf_counter() - t0
        kernel.set_tile_culling(None)
        if plain_screen is not None:
            culled_same = bool(torch.equal(plain_screen, screen))

    stats = torch.tensor([dt, float(prof.rays_primary), float(prof.rays_shadow), float(prof.sweep_tests),
                          float(prof.shadow_tests), prof.sweep_ms, float(prof.sweep_launches),
                          prof.shadow_ms, float(prof.exact_candidates), float(prof.exact_overflow), dt_culled],
                         dtype=torch.float64, device=dev)
    tmax = stats.clone()
    tsum = stats.clone()
    # every rank's own clock and frame count (a sequence is split by frames: rank r renders frames r, r + world, ...)
    mine = torch.tensor([dt, float(args.steps)], dtype=torch.float64, device=dev)
    per_rank = [mine.clone() for _ in range(world)] if use_dist else [mine]
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dist.all_gather(per_rank, mine)
    if rank == 0:
        wall = tmax[0].item()
        steps = args.steps
        # ray and pair counts: the profiled frame, summed over ranks
        rays_p, rays_s = tsum[1].item(), tsum[2].item()
        tests = tsum[3].item() + tsum[4].item()
        hit_frac = float((out["hit"] >= 0).float().mean().item())
        # roofline of the dominant kernel: rank 0's closest-hit sweep launch of the profiled frame
        sweep_ms_avg = prof.sweep_ms / max(prof.sweep_launches, 1)
        tests_per_launch = prof.sweep_tests / max(prof.sweep_launches, 1)
        f_exec = f_primary.value                  # flops the hot loop really executes per pair
        mix_ppl, n_fma, n_add, n_logic, n_cmp = [m.value for m in mix]
        instr_per_test = (n_fma + n_add + n_logic + n_cmp) / mix_ppl
        tests_per_s_launch = tests_per_launch / (sweep_ms_avg * 1e-3) if sweep_ms_avg > 0 else 0.0
        achieved = tests_per_s_launch * f_exec / 1e12
        traffic = measured_traffic(args.workload, f"raster-tile{tile.value}-ppl{ppl.value}-chunks{chunks.value}") if world == 1 else None
        ms_step = wall / steps * 1e3
        # (4096 pixels per lane and more: the wave-block form, 4096 columns by 64 NL rows per workgroup)
        raster_name = (f"k_sweep_raster_wave<{ppl.value // 4096}>" if ppl.value >= 4096 else
                       f"k_sweep_raster<{ppl.value}, false, {'true' if regular.value else 'false'}>")
        rp = rocprof_avg_ms(args.workload, raster_name, minus_fastest=hot_only_ms is not None) if world == 1 else None
        # the any-hit sweeps of the shadow rays
        sh_ms, sh_tests = prof.shadow_ms, float(prof.shadow_tests)
        sh_rate = sh_tests / (sh_ms * 1e-3) if sh_ms > 0 else 0.0
        if prof.cell_launches > 0:
            # Sweeps over lattices of direction cells.  The hot loop decides every (cell, triangle) pair
            # of a wave's footprint through a separable test -- an addition per column and per row of
            # the footprint (dealt to the lanes), a three-way minimum per two of either, two compares.  The library keeps the lattice
            # coarse (a few footprints per set of rays), so that loop only tells which triangles come
            # near a wave's footprint; behind it the rays of the fine cells a triangle's square
            # touches are looked up (a prefix table over the sorted rays) and tested one by one,
            # and that part -- dependent gathers, not arithmetic -- is what the launch's time is.
            cpl = int(prof.cells_per_lane)
            # (a wave's footprint is 8 RX columns by 8 RY rows of cells, dealt to its 64 lanes: at least
            # one column and one row test per lane)
            c_ry = {2048: 32, 512: 16, 128: 8, 32: 4, 8: 2}[cpl]
            c_cols, c_rows = max(8 * (cpl // c_ry) // 64, 1), max(8 * c_ry // 64, 1)
            c_instr = (c_cols + c_rows + c_cols // 2 + c_rows // 2 + 2) / cpl
            c_flop = (c_cols + c_rows) / cpl
            c_rate = prof.cell_pairs / (prof.cell_ms * 1e-3) if prof.cell_ms > 0 else 0.0
            one_by_one = float(prof.cell_ray_pairs)
            rp_sh = rocprof_avg_ms(args.workload, f"k_sweep_cells<{cpl}>") if world == 1 else None
            anyhit = {
                "kernel": f"k_sweep_cells<{cpl}> (shadow rays, any hit: every cell of a coarse lattice of directions "
                          "against every triangle, then the rays of the fine cells a triangle's square touches one by one)",
                "bound": "latency of the rare path's gathers (row look-ups, rays, candidate records); the hot loop is "
                         "a small part of the launch at this lattice size",
                "ms_per_frame": sh_ms, "share_of_step": sh_ms / ms_step, "launches_per_frame": int(prof.shadow_launches),
                "lattice_launches": int(prof.cell_launches), "ms_in_lattice_launches": prof.cell_ms,
                "rays_swept": float(prof.cell_rays), "cell_pairs_per_frame": float(prof.cell_pairs),
                "cell_pairs_per_s": c_rate,
                "fine_cells_per_cell": int(prof.cell_fine),
                "triangles_entering_rare_path_per_frame": float(prof.cell_entries),
                "row_lookups_per_frame": float(prof.cell_lookups),
                "ray_pairs_one_by_one_per_frame": one_by_one,
                "ray_pairs_one_by_one_per_s": one_by_one / (prof.cell_ms * 1e-3) if prof.cell_ms > 0 else 0.0,
                "tests_per_frame": sh_tests, "tests_per_s": sh_rate,
                "test": "a (ray, triangle) pair decided: every listed ray against every triangle, most pairs through "
                        "their cell's test or through lying outside the fine cells a triangle's square touches",
                "flop_per_cell_pair": c_flop, "achieved": c_rate * c_flop / 1e12, "peak": PEAK_FP32_TFLOPS,
                "unit": "TFLOP/s", "frac": c_rate * c_flop / 1e12 / PEAK_FP32_TFLOPS,
                "rocprof": rp_sh,
                "valu_issue": {"instr_per_cell_pair": c_instr,
                               "per_lane_and_triangle": {"cells": cpl, "add": c_cols + c_rows,
                                                         "logic": c_cols // 2 + c_rows // 2, "compare": 2},
                               "achieved": c_rate * c_instr / 1e12, "peak": PEAK_VALU_TINSTR, "unit": "T lane-instr/s",
                               "frac": c_rate * c_instr / 1e12 / PEAK_VALU_TINSTR,
                               "note": "hot loop only; see `bound`"},
                "cells_per_lane": cpl,
            }
        else:
            # hot loop of the widest ray-by-ray kernel (u half of the bounding-square test): a subtraction
            # per ray, a three-way minimum per two, one compare per triangle (25 VALU instructions per 16 pairs)
            sh_instr = (spl.value + spl.value // 2 + 1) / spl.value
            rp_sh = rocprof_avg_ms(args.workload, f"k_sweep_apex<{spl.value}, false>") if world == 1 else None
            anyhit = {
                "kernel": f"k_sweep_apex<{spl.value}, false> (shadow rays, any hit, ray by ray)",
                "ms_per_frame": sh_ms, "share_of_step": sh_ms / ms_step, "launches_per_frame": int(prof.shadow_launches),
                "tests_per_frame": sh_tests, "tests_per_s": sh_rate,
                "flop_per_test": f_shadow.value,
                "achieved": sh_rate * f_shadow.value / 1e12, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": sh_rate * f_shadow.value / 1e12 / PEAK_FP32_TFLOPS,
                "rocprof": rp_sh,
                "valu_issue": {"instr_per_test": sh_instr, "achieved": sh_rate * sh_instr / 1e12,
                               "peak": PEAK_VALU_TINSTR, "unit": "T lane-instr/s",
                               "frac": sh_rate * sh_instr / 1e12 / PEAK_VALU_TINSTR,
                               **mix_alone(f"apex{spl.value}_strip", sh_rate)},
                "rays_per_lane": spl.value,
            }
        primary_ms = prof.sweep_ms
        comm = {"backend": (dist.get_backend() if use_dist else None),
                "world_size": (dist.get_world_size() if use_dist else 1),
                "exchange": ("none: every rank renders whole frames" if frame_parallel else
                             ("gather of the 8-bit rows to rank 0" if use_dist else "none: one process, one device")),
                "gather_ms": gather_ms[0],
                "gather_ms_is": "exchange + placement of the profiled frame on rank 0, host clock, all ranks idle before it"}
        if single:
            comm = {"backend": ms.exchange, "world_size": 1, "devices": devices,
                    "exchange": "rt_render_frame_multi: rows to the first device inside the call", "gather_ms": None}
        res = {
            "metric": "Mrays/s and ray-tri tests/s (% fp32 peak) at 1/2/4/8 MI355X vs host-CPU ref",
            "value": (rays_p + rays_s) / (wall / steps) / 1e6,
            "unit": "Mrays/s",
            "n_gpus": len(devices) if single else world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": wall / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if frame_parallel else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": args.workload, "mesh": mesh.name, "mesh_source": mesh_note, "triangles": mesh.num_tris,
                       "width": W, "height": H, "lights": frame.num_lights,
                       "sharding": f"{world} ranks, one whole frame each per step (frames rank, rank + {world}, ...), no exchange"
                       if frame_parallel else (f"{world} interleaved row parts (16-row blocks), "
                                    f"{'RCCL' if args.backend == 'nccl' else 'gloo (rehearsal)'} gather to rank 0")
                       if world > 1 else (f"one process, devices {devices}: interleaved row parts, rt_render_frame_multi "
                                          "(one host thread per device, rows to the first device over RCCL send/recv; peer copies where a device is named twice)"
                                          if single else "single GPU")},
            "frames_per_s": steps * (world if frame_parallel else 1) / wall,
            # whole frames each rank rendered in the timed region and its own seconds (a sequence under
            # several ranks: the frame rate above is the sum of the ranks' frames over the slowest rank's clock)
            "frames": {"per_rank": [int(t[1].item()) for t in per_rank] if frame_parallel else [steps] * (world if use_dist else 1),
                       "whole_frames_per_rank": bool(frame_parallel or world == 1),
                       "seconds_per_rank": [t[0].item() for t in per_rank]},
            "primary_mrays_per_s": rays_p / (wall / steps) / 1e6,
            "rays_per_frame": {"primary": rays_p, "shadow": rays_s},
            "tests_per_s": tests / (wall / steps),
            "tests_per_s_counts": "(ray, triangle) pairs decided per second of the frame: every primary ray and every "
                                  "swept shadow ray against every triangle; the sweeps decide most pairs through a box "
                                  "test shared by a lane's lattice, not one by one (roofline.general_mt is the sweep "
                                  "that evaluates each pair on its own)",
            "comm": comm,
            "hit_fraction_rank0": hit_frac,
            "fp64_confirmations_per_frame": tsum[8].item(),
            "overflow_rays_per_frame": tsum[9].item(),
            "timing": "steps timed with the library's profiling off; launch times, ray and pair counts from one "
                      "more frame rendered with it on",
            # Roofline of the dominant kernel, the closest-hit sweep of the primary rays.  It is a
            # vector-ALU kernel: `achieved` counts the fp32 flops its hot loop executes per
            # ray-triangle pair (an addition = 1; rocprof SQ_INSTS_VALU_{ADD,MUL,FMA}_F32 agree,
            # see profiles/) against the fp32 vector peak.  What actually bounds the loop is
            # instruction issue (one VALU instruction per SIMD every two cycles, whatever it
            # computes): the pixels of a lane (lattice form) or of a wave's block (wave-block form,
            # 4096 pixels per lane and more) are rows times columns, and per triangle the loop
            # steps each column's and each row's distance from the centre of the triangle's
            # bounding box by one addition or subtraction and collects two of those in a three-way
            # minimum (a pixel is in the box when its column and its row are), so `valu_issue` is
            # the figure that says how close the hot loop is to the machine's limit, and the stated
            # target of half the fp32 flop peak is not met by construction.  With the tall blocks
            # of the wave-block form that loop is a small part of the launch (`hot_loop_only_ms`);
            # the rest is the rare path, which works through the pixel rectangles of the triangles
            # that passed, bound by memory latency, not by arithmetic (`bound` says which it is).
            # The pairs are those of the image's own pixels: columns beyond the right edge and
            # rows beyond the bottom are tested too, uncounted.
            "roofline": {
                "bound": "valu-fp32" if hot_only_ms is None or hot_only_ms >= 0.5 * sweep_ms_avg else
                         "latency of the rare path (the hot loop alone: hot_loop_only_ms; the figures below are the hot loop's)",
                "hot_loop_only_ms": hot_only_ms,
                "hot_loop_share_of_launch": (hot_only_ms / sweep_ms_avg) if hot_only_ms and sweep_ms_avg > 0 else None,
                # the launch with the rare path off against the hot loop's instruction sequence alone
                "hot_loop_frac_of_mix_alone": mix_alone(f"raster{mix_ppl}_wave" if mix_ppl >= 4096 else f"raster{mix_ppl}_box",
                                                        tests_per_launch / (hot_only_ms * 1e-3)).get("frac_of_mix_alone")
                                              if hot_only_ms else None,
                "kernel": f"{raster_name} (primary rays, closest hit, all pairs)",
                "dominant_by_time": "this kernel" if primary_ms >= sh_ms else "anyhit (below)",
                "share_of_step": primary_ms / ms_step,
                "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_TFLOPS,
                "target_50pct_fp32_flops_met": bool(achieved / PEAK_FP32_TFLOPS >= 0.5),
                "avg_launch_ms": sweep_ms_avg, "tests_per_launch": tests_per_launch,
                "tests_per_s": tests_per_s_launch,
                "flop_per_test": f_exec,
                # the same fraction from the committed rocprofv3 summary's average duration
                "frac_rocprof": (tests_per_launch * f_exec / (rp["avg_ms"] * 1e-3) / 1e12 / PEAK_FP32_TFLOPS) if rp else None,
                "rocprof": rp,
                "valu_issue": {"instr_per_test": instr_per_test,
                               "per_lane_and_triangle": {"pixels": mix_ppl, "fma": n_fma, "add": n_add,
                                                         "logic": n_logic, "compare": n_cmp},
                               "achieved": tests_per_s_launch * instr_per_test / 1e12,
                               "peak": PEAK_VALU_TINSTR, "unit": "T lane-instr/s",
                               "frac": tests_per_s_launch * instr_per_test / 1e12 / PEAK_VALU_TINSTR,
                               # the same instruction sequence with nothing around it (no LDS reads,
                               # barriers, branches or rare path), measured by tools/microbench/valu_rate
                               **mix_alone(f"raster{mix_ppl}_wave" if mix_ppl >= 4096 else f"raster{mix_ppl}_box", tests_per_s_launch)},
                # every vector instruction of the launch (hot loop and rare path), counted by rocprofv3,
                # against the issue peak over this run's launch time: how busy the vector unit is
                "valu_issue_whole_launch": (lambda c: {**c, "frac": c["wave_instructions"] * 64 / (sweep_ms_avg * 1e-3) / 1e12 / PEAK_VALU_TINSTR}
                                            if c and sweep_ms_avg > 0 else None)(counted_valu(args.workload, raster_name)),
                "traffic": traffic["bytes"] if traffic else None,
                "traffic_source": traffic["source"] if traffic else None,
                "sweep_kernel": {"tile_tris": tile.value, "pixels_per_lane": ppl.value,
                                 "triangle_chunks": chunks.value},
                # second kernel: the any-hit sweeps (rank 0, all launches of the profiled frame)
                "anyhit": anyhit,
            },
        }
        if world == 1 and not single and not sequence and not args.no_general_mt:
            # the literal Moller-Trumbore kernel on the same rays, and on the 100 k-triangle mesh
            res["roofline"]["general_mt"] = general_mt(sc, W, H, dev, out["hit"])
            if args.workload == "soup1m_4k" and obj is None:
                k2, t2, _, _ = WORKLOADS["mesh100k_4k"]
                sc2 = kernel.Scene(make_mesh(k2, t2), device=local_rank)
                sc2.transform(obmats[0])
                res["roofline"]["general_mt_mesh100k_4k"] = general_mt(sc2, W, H, dev)
                sc2.close()
        if culling_was_on:
            res["config"]["tile_culling"] = "on for the whole run (RT_AMD_CULL): the figures above are not the all-pairs sweep's"
        else:
            wall_c = tmax[10].item()
            res["tile_culling"] = {
                "what": "the same steps with rt_set_tile_culling(1): a tile's triangles are tested against a wave's "
                        "whole ray bundle before the per-ray tests; off by default, not part of `value`",
                "ms_per_step": wall_c / steps * 1e3,
                "value": (rays_p + rays_s) / (wall_c / steps) / 1e6, "unit": "Mrays/s",
                "frames_per_s": steps * (world if frame_parallel else 1) / wall_c,
                "speedup_vs_all_pairs": wall / wall_c if wall_c > 0 else None,
                "screen_identical_to_all_pairs": culled_same,
            }
        if single:
            # the picture the devices assembled against the profiled single-scene render of the same frame
            res["verify"] = {"screen_equals_single_rank_render": bool(torch.equal(single_screen, screen))}
        if args.verify and not single:
            k = ((step_no[0] - 1) * world + rank if frame_parallel else step_no[0] - 1) % len(frames)
            sc.transform(obmats[k])
            alone = sc.render_rows(frames[k], W, H, TAN_HALF, want=("rgb8",))["rgb8"]
            res["verify"] = {"screen_equals_single_rank_render":
                             bool(torch.equal(sc.place_rows(alone, W, H, 0, H), screen))}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(mesh, frame, W, H, args.cpu_budget)
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
