#!/usr/bin/env python3
"""Benchmark of the MI355X ORB front end.

A step = one pass of the extractor over one batch of synthetic KITTI-shape frames (1241x376,
2000 keypoints per frame, 8 levels, scale 1.2, FAST thresholds 20/7) that already sit in HBM.
`value` is frames per second over all ranks.  One process per GPU; frames are independent, so
ranks share nothing on the data path (weak scaling); torch.distributed only provides the barrier
and the max-over-ranks of the elapsed time.

After the timed region one extractor instance runs the same launches alone with stage profiling on
(the library then keeps every stage on one stream), which gives exclusive per-kernel durations.
`--exclusive` runs the timed region itself that way (one instance, no overlap), so that a
`rocprofv3 --kernel-trace --stats` of that command shows the same per-kernel averages.

Extra objects on the JSON line:
  roofline      dominant extractor kernel: algorithmic bytes per launch / its exclusive HIP-event duration
                vs 8 TB/s; `traffic` = HBM bytes per launch from committed PMC passes (profiles/), if any
  cpu_baseline  the CPU oracle (oracle/, a scalar restatement of the reference) on a bounded sample
  stages        per-stage milliseconds per batch (HIP events on the extractor's stream)
  hamming       brute-force top-2 of 2000 queries against a 1M-descriptor database on this GPU
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
# Vector-ALU issue rates measured by tools/ubench_valu.hip (profiles/r02/ubench_valu.txt), chip-wide, in
# wave-instructions per second with 8 waves per SIMD: packed 16-bit min/max/add, byte permute / align, 24-bit
# multiply-add, dot products, popcount, 3-input and compare / select instructions issue at 0.59e12 (one per 4
# clocks and SIMD); 32-bit add / sub / and / or / xor / shift right, f32 add / mul / fma and moves at about 1.0e12.
VALU_WAVE_INSTR_PER_S_SLOW = 0.59e12
VALU_WAVE_INSTR_PER_S_FAST = 1.0e12
# INT32 lane rate that the popcount roofline of the matcher is quoted on: 256 CUs x 64 lanes per clock x 2.4 GHz;
# one 256-bit pair = 8 XOR + 8 popcount-accumulate
VALU_LANE_OPS_PER_S = 256 * 64 * 2.4e9
OPS_PER_PAIR = 16
# The top-2 kernel runs on the matrix cores for 32 or more queries: a pair is 256 FP4 multiply-adds of
# v_mfma_scale_f32_32x32x64_f8f6f4 (dense FP4 peak: twice the fp8 rate = 2048 MAC per clock and SIMD, 1024 SIMDs,
# 2.4 GHz); tools/probe_mfma_f4.hip sustains 3.74e15 MAC/s with random +-1 operands in registers
# (the i8 products the kernel used before: peak 2.5e15, 2.12e15 sustained, tools/ubench_mfma_i8.hip)
MFMA_F4_MAC_PER_S = 2048 * 1024 * 2.4e9
MFMA_F4_MAC_MEASURED = 3.737e15
MAC_PER_PAIR = 256

WORKLOADS = {
    # name: (width, height, nfeatures, frames per batch)
    # 1024 frames per step = 256 per launch of each of the four instances (128 per launch: 2 % fewer frames/s,
    # 64 per launch: 5 % fewer)
    "kitti_1241x376_2000kp": (1241, 376, 2000, 1024),
    "euroc_752x480_1000kp": (752, 480, 1000, 128),
    "tum_640x480_1000kp": (640, 480, 1000, 128),
}


def level_sizes(w, h, nlevels=8, scale=1.2):
    s = np.float32(1.0)
    out = []
    for _ in range(nlevels):
        inv = np.float32(1.0) / s
        out.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
        s = np.float32(np.float64(s) * np.float64(np.float32(scale)))
    return out


def algorithmic_bytes(w, h, n_kp):
    """Compulsory bytes per frame of each stage (every stage reads its input once and writes its
    output once): P = pyramid pixels, I = level-0 pixels, p7 = last-level pixels, N = keypoints."""
    lv = level_sizes(w, h)
    P = sum(a * b for a, b in lv)
    I = lv[0][0] * lv[0][1]
    p7 = lv[-1][0] * lv[-1][1]
    per_stage = {
        "ingest": 0.0,
        "resize": float((P - p7) + (P - I)),
        "fast": float(P),
        "octree": 0.0,
        "blur": float(2 * P),
        "describe": float((749 + 512 + 60) * n_kp),
    }
    return per_stage, float(5 * P - I - p7 + 1321 * n_kp)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="kitti_1241x376_2000kp", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="frames per step per GPU (default: workload's)")
    ap.add_argument("--instances", type=int, default=4, help="extractor instances (streams) sharing a step's frames")
    ap.add_argument("--exclusive", action="store_true",
                    help="one instance, stages serialised and profiled inside the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hamming", action="store_true")
    ap.add_argument("--frames-resident", type=int, default=2048,
                    help="synthetic frames resident in HBM per GPU; the steps rotate through them (default: a set larger "
                         "than the 256 MiB Infinity Cache, so that level 0 cannot stay cached between steps)")
    ap.add_argument("--no-config5", action="store_true")
    ap.add_argument("--no-pcie", action="store_true")
    ap.add_argument("--pcie-only", action="store_true", help="only the host-memory-to-host-memory measurement")
    ap.add_argument("--pcie-instances", type=int, default=2)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-pipeline", action="store_true",
                    help="the extractor instances run on independent streams instead of taking turns in one "
                         "orbx_pipeline (comparison runs)")
    ap.add_argument("--hw-queues", type=int, default=8,
                    help="GPU_MAX_HW_QUEUES for this process unless the environment already asks for more (HIP's "
                         "default of 4 makes the instances' streams share hardware queues); 0: leave it")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: the ranks only form the process group, pass the barriers and the reductions of "
                         "the timed region and rank 0 prints a line with value null (checks the launch on a CPU box)")
    args = ap.parse_args()
    # before anything starts the HIP runtime (the ranks started below inherit it)
    env_before = dict(os.environ)
    if args.hw_queues and int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < args.hw_queues:
        os.environ["GPU_MAX_HW_QUEUES"] = str(args.hw_queues)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Called without a launcher: this process has not touched the GPU (nothing above imports torch), so it
        # may start the ranks as children.  It never becomes a rank itself.
        raise SystemExit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # a line that says n_gpus = N must come from N ranks
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); refusing to run\n"
                         % (args.gpus, world))
        raise SystemExit(2)
    # ORBX_BENCH_FORCE_DIST=1: take the multi-rank code path (process group, barriers, all-gather) even
    # with one rank, to rehearse it with the nccl backend on a one-GPU box
    dist_path = world > 1 or os.environ.get("ORBX_BENCH_FORCE_DIST") == "1"
    backend = os.environ.get("ORBX_BENCH_BACKEND", "nccl")  # "gloo" rehearses the N>1 path on one GPU
    share_gpu = os.environ.get("ORBX_BENCH_SHARE_GPU") == "1"  # rehearsal: more ranks than GPUs, ranks share devices
    # The host-memory measurement (rank 0's own business) runs in a process of its own, started here, before
    # this one touches the GPU: its two instances want the streams of a fresh process on HIP's default number of
    # hardware queues (104 k frames/s; 46-66 k inside this process, after or beside the streams of the main
    # measurement or with 8 queues).  The other ranks meanwhile wait at the first barrier.
    pcie_first = None
    if not args.no_pcie and not args.pcie_only and not args.dry_run and rank == 0:
        pcie_first = pcie_in_child(args, env_before, local_rank)
    import torch
    import torch.distributed as dist

    if args.dry_run:
        return dry_run(args, torch, dist, rank, world, dist_path)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the only path")
    ndev = torch.cuda.device_count()
    if ndev < world and not (share_gpu and backend != "nccl"):
        sys.stderr.write("bench.py: %d ranks but %d GPU(s) visible; one rank per GPU (a rehearsal with ranks sharing a "
                         "GPU needs ORBX_BENCH_SHARE_GPU=1 and ORBX_BENCH_BACKEND=gloo)\n" % (world, ndev))
        raise SystemExit(2)
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    if dist_path:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            sys.stderr.write("bench.py: process group of %d ranks for --gpus %d\n" % (dist.get_world_size(), args.gpus))
            raise SystemExit(2)

    import orbslam2_amd as orbx

    w, h, nfeat, batch = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    # synthetic frames, different on every rank, generated on the host and uploaded once
    stride = (w + 63) // 64 * 64  # pitch-aligned rows: the extractor reads the frames in place
    nres = max(batch, args.frames_resident // batch * batch)
    host = np.zeros((nres, h, stride), np.uint8)

    def synth(f):
        host[f] = orbx.synth_image(0x0B5EED00 + rank * 100003 + f, w, h, 0, stride=stride)

    import concurrent.futures
    with concurrent.futures.ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as pool:
        list(pool.map(synth, range(nres)))
    if args.pcie_only:
        print(json.dumps(pcie_inclusive(orbx, torch, host, w, h, nfeat, local_rank, instances=args.pcie_instances)))
        return
    frames = torch.from_numpy(host).cuda()
    # A stereo pair batch: left and right images go through separate extractor instances on separate
    # streams, as the reference runs mpORBextractorLeft / mpORBextractorRight in two threads
    # (src/Frame.cc:79-82).  The step's frames are split evenly over the instances (default 4: two
    # left/right pairs of extractors), whose kernels the GPU overlaps.
    ninst = 1 if args.exclusive else max(1, args.instances)
    half = (batch + ninst - 1) // ninst
    streams = [torch.cuda.Stream() for _ in range(ninst)]
    exs = [orbx.ORBextractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=half, device=local_rank)
           for _ in range(ninst)]
    # the instances take turns: each one's blur runs behind the work of the instance called before it
    pipeline = None if (args.exclusive or args.no_pipeline or ninst < 2) else orbx.Pipeline(local_rank)
    for e, st in zip(exs, streams):
        e.set_stream(st.cuda_stream)
        if pipeline is not None:
            e.set_pipeline(pipeline)
    stream = streams[0]
    ex = exs[0]
    nwin = nres // batch  # windows of `batch` frames; step s works on window s % nwin
    eyes = [(frames[i * half:(i + 1) * half], min(half, batch - i * half)) for i in range(ninst)]
    fbytes = h * stride
    counter = [0]

    def step():
        base = (counter[0] % nwin) * batch * fbytes
        counter[0] += 1
        for e, (fr, nb) in zip(exs, eyes):
            e.extract_batch_device(fr.data_ptr() + base, nb, w, h, stride, fbytes)

    def barrier():
        torch.cuda.synchronize()
        if dist_path:
            dist.barrier()
        torch.cuda.synchronize()

    if args.exclusive:
        ex.profile(True)
    for _ in range(args.warmup):
        step()
    barrier()
    if args.exclusive:
        ex.profile(True)  # drops the warm-up samples
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    el = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
    if dist_path:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    n_kp = float(np.mean(np.concatenate([e.batch_counts()[0] for e in exs])))
    # exclusive per-stage times: instance 0 alone, same launch size, stages serialised by the library
    if not args.exclusive:
        ex.profile(True)
        fr0, nb0 = eyes[0]
        for _ in range(max(args.steps, 5)):
            ex.extract_batch_device(fr0.data_ptr(), nb0, w, h, stride, h * stride)
        torch.cuda.synchronize()
    stage_ms, nprof = ex.stage_times()
    ex.profile(False)
    stage_ms = {k: v / max(nprof, 1) for k, v in stage_ms.items()}
    launch_frames = eyes[0][1]

    if rank != 0:
        # the measurements with a collective need every rank; a rank that cannot finish one leaves after the
        # same limit as rank 0, so that no rank waits in a collective forever
        if not args.no_config5:
            guarded(lambda: bench_config5(orbx, torch, dist, rank, world, w, h, nfeat, batch, stride, coll_dev), None,
                    "config5", True)
        if dist_path and not args.no_hamming:
            guarded(lambda: bench_hamming_sharded(orbx, torch, dist, stream, rank, world, coll_dev), None, "hamming", True)
        if dist_path:
            dist.destroy_process_group()
        return

    frames_total = world * batch * args.steps
    fps = frames_total / elapsed
    per_stage_bytes, frame_bytes = algorithmic_bytes(w, h, n_kp)
    # the dominant kernel: stages that are one kernel launch (resize is a chain of 7 smaller launches;
    # "octree" is the distribution kernel plus a ~20 us compaction kernel and moves almost no data)
    single = {"fast": "k_fast", "blur": "k_blur", "describe": "k_describe"}
    dom = max(single, key=lambda k: stage_ms[k])
    dom_bytes = per_stage_bytes[dom] * launch_frames
    dom_s = stage_ms[dom] * 1e-3
    achieved = dom_bytes / dom_s / 1e9 if dom_s > 0 else 0.0
    traffic, traffic_src = committed_traffic(single[dom], args.workload, launch_frames)
    roofline = {"bound": "hbm", "kernel": single[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel_ms_per_launch": round(stage_ms[dom], 4),
                "algorithmic_bytes_per_launch": dom_bytes,
                "limiter": "vector-ALU issue and LDS / barrier latency, not HBM: see valu_* fields",
                "pipeline_achieved_GBs": round(fps / world * frame_bytes / 1e9, 2),
                "pipeline_frac": round(fps / world * frame_bytes / 1e9 / HBM_PEAK_GBS, 5)}
    valu = committed_valu(single[dom], args.workload, launch_frames)
    if valu is not None and dom_s > 0:
        roofline["valu_wave_instr_per_launch"] = valu
        roofline["valu_wave_instr_per_s"] = round(valu / dom_s / 1e12, 4)
        roofline["valu_issue_ceilings_T_wave_instr_per_s"] = {
            "packed_16bit_perm_mad_dot_3input": VALU_WAVE_INSTR_PER_S_SLOW / 1e12,
            "add_sub_logic_f32": VALU_WAVE_INSTR_PER_S_FAST / 1e12,
            # measured streams that alternate the two classes; the kernel's own mix lies between the two
            "mix_1_fast_per_slow": 0.65, "mix_3_fast_per_slow": 0.77,
            "source": "tools/ubench_valu.hip, profiles/r02/ubench_valu.txt"}
        roofline["valu_frac_of_slow_class_ceiling"] = round(valu / dom_s / VALU_WAVE_INSTR_PER_S_SLOW, 4)

    out = {
        "metric": "ORB frames/s @KITTI 1241x376 2000 kp (1/8 GPU); Hamming matches/s; % roofline",
        "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "ranks": dist.get_world_size() if dist_path else 1,
        "backend": backend if dist_path else None, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": args.workload, "frames_per_step_per_gpu": batch, "extractor_instances": ninst,
                   "frames_resident_per_gpu": nres, "frames_resident_MiB": round(nres * fbytes / 2**20, 1),
                   "working_set": "steps rotate through the resident frames; a step also writes and reads %.0f MiB of "
                                  "pyramid and blurred levels" % (batch * 2 * (per_stage_bytes["fast"]) / 2**20),
                   "nfeatures": nfeat, "nlevels": 8,
                   "scale_factor": 1.2, "ini_th_fast": 20, "min_th_fast": 7, "mean_keypoints_per_frame": n_kp,
                   "parallelism": "frame-batch shards, no collective"},
        "roofline": roofline,
        "stages_ms_per_launch_exclusive": {k: round(v, 4) for k, v in stage_ms.items()},
        "frames_per_launch": launch_frames,
        "timed_region": "exclusive (1 instance, serialised stages)" if args.exclusive else
                        ("%d instances taking turns in one orbx_pipeline: one stream each, an instance's blur behind "
                         "the work of the instance before it" % ninst) if pipeline is not None else
                        "%d instances on %d independent streams, blur overlapped" % (ninst, 2 * ninst),
        "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
    }

    if not args.no_config5:
        out["config5"] = guarded(lambda: bench_config5(orbx, torch, dist if dist_path else None, rank, world, w, h, nfeat,
                                                       batch, stride, coll_dev), out, "config5", dist_path)
    if not dist_path and not args.no_hamming:
        out["hamming"] = bench_hamming(orbx, torch, stream)
    if dist_path and not args.no_hamming:
        # the frames/s line must survive a failure or a hang of this extra measurement
        out["hamming"] = guarded(lambda: bench_hamming_sharded(orbx, torch, dist, stream, rank, world, coll_dev), out,
                                 "hamming", True)
    if dist_path:
        dist.destroy_process_group()  # what follows is rank 0's own business
    del frames
    for e in exs:
        e.close()  # their streams would share the few hardware queues with those of the next measurement
    if not args.no_pcie:
        out["pcie_inclusive"] = pcie_first if pcie_first is not None else pcie_inclusive(
            orbx, torch, host, w, h, nfeat, local_rank, instances=args.pcie_instances)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(host, w, nfeat, args.cpu_seconds)
        out["speedup_vs_cpu_baseline"] = round(fps / world / out["cpu_baseline"]["value"], 1)
        out["speedup_vs_cpu_baseline_what"] = "frames/s of one GPU over frames/s of one host core"
    print(json.dumps(out), flush=True)


def launch_ranks(n):
    """Starts `n` ranks of this script (one per GPU) through torch.distributed.run as a child process, relays
    their output (rank 0 prints the JSON line) and returns the child's exit status, which is non-zero if any
    rank failed.  The parent stays off the GPU and is not replaced: it waits."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(args, torch, dist, rank, world, dist_path):
    """The launch and the collectives of the timed region without any GPU work (gloo)."""
    if dist_path:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        if dist.get_world_size() != args.gpus:
            raise SystemExit(2)
        dist.barrier()
    t0 = time.perf_counter()
    el = torch.tensor([time.perf_counter() - t0 + 1e-6 * (rank + 1)], dtype=torch.float64)
    if dist_path:
        dist.barrier()
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    ranks = dist.get_world_size() if dist_path else 1
    if dist_path:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "ORB frames/s @KITTI 1241x376 2000 kp (1/8 GPU); Hamming matches/s; % roofline",
                          "value": None, "unit": "frames/s", "n_gpus": world, "ranks": ranks,
                          "backend": "gloo" if dist_path else None, "steps": args.steps, "warmup": args.warmup,
                          "dry_run": True, "max_over_ranks_s": float(el.item())}), flush=True)


SHARDED_HAMMING_LIMIT_S = 150


def guarded(fn, out, key, multi_rank):
    """Runs an extra measurement under a time limit.  An exception becomes an error entry.  At the limit (a
    rank stuck in a collective because another rank failed) rank 0 prints the line it has (`out`, None on
    other ranks) with an error entry, and every rank leaves the process with a non-zero status: a hang must
    not look like a successful run."""
    import threading

    def bail():
        if out is not None:
            out[key] = {"error": "measurement did not finish in %d s" % SHARDED_HAMMING_LIMIT_S}
            print(json.dumps(out), flush=True)
        os._exit(3)

    timer = threading.Timer(SHARDED_HAMMING_LIMIT_S, bail) if multi_rank else None
    if timer:
        timer.daemon = True
        timer.start()
    try:
        return fn()
    except Exception as exc:
        return {"error": repr(exc)}
    finally:
        if timer:
            timer.cancel()


def _committed(name):
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        return json.load(fh)


def committed_traffic(kernel, workload, launch_frames):
    """HBM bytes per launch of `kernel` from the committed counter passes (profiles/traffic.json holds
    per-frame bytes measured with TCC FETCH_SIZE / WRITE_SIZE at the same launch size); None if absent."""
    t = _committed("traffic.json")
    if not t or t.get("workload") != workload or kernel not in t.get("kernels", {}):
        return None, None
    k = t["kernels"][kernel]
    return round(k["hbm_bytes_per_frame"] * launch_frames, 1), t.get("source")


def committed_valu(kernel, workload, launch_frames):
    t = _committed("traffic.json")
    if not t or t.get("workload") != workload or kernel not in t.get("kernels", {}):
        return None
    v = t["kernels"][kernel].get("valu_wave_instr_per_frame")
    return None if v is None else v * launch_frames


def config4_data(orbx, nq, ndb):
    """Relocalisation-scale matching data: a random database (splitmix64 seed 1); query i = database row
    i*(ndb/nq) with 0..80 distinct bits flipped (seed 2); next to every tenth source row sits a
    near-duplicate (4 further flipped bits), so that best and second best are close and the ratio test fails."""
    db = orbx.synth_descriptors(1, ndb)
    rng = np.random.default_rng(2)
    step = ndb // nq
    rows = np.arange(nq, dtype=np.int64) * step
    bits = np.unpackbits(db[rows], axis=1)
    for i in range(nq):
        bits[i, rng.choice(256, size=int(rng.integers(0, 81)), replace=False)] ^= 1
        if i % 10 == 0:
            b2 = np.unpackbits(db[rows[i]])
            b2[rng.choice(256, size=4, replace=False)] ^= 1
            db[rows[i] + 1] = np.packbits(b2)
    return db, np.packbits(bits, axis=1), rows


def numpy_top2(q, db):
    """Single pass of the reference's bookkeeping over all rows for one query, vectorised."""
    d = np.bitwise_count(db.view(np.uint64) ^ q.view(np.uint64)).sum(axis=1).astype(np.int64)
    key = d * (1 << 32) + np.arange(len(db))
    two = np.partition(key, 1)[:2]
    two.sort()
    return int(two[0] & 0xFFFFFFFF), int(two[0] >> 32), int(two[1] >> 32)


def bench_hamming(orbx, torch, stream):
    """2000 queries x 1M database rows resident in HBM (one GPU's view of the sharded relocalisation match)."""
    nq, ndb = 2000, 1_000_000
    hdb, hq, _ = config4_data(orbx, nq, ndb)
    db = torch.from_numpy(hdb).cuda()
    q = torch.from_numpy(hq).cuda()
    best = torch.zeros(nq, dtype=torch.int32, device="cuda")
    d1 = torch.zeros(nq, dtype=torch.int16, device="cuda")
    d2 = torch.zeros(nq, dtype=torch.int16, device="cuda")
    m = orbx.ORBmatcher(0.75, device=torch.cuda.current_device())
    m.set_stream(stream.cuda_stream)

    def run():
        m.match_top2_device(q.data_ptr(), nq, db.data_ptr(), ndb, 0, best.data_ptr(), d1.data_ptr(), d2.data_ptr())

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    reps = 20
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        run()
    e1.record(stream)
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / reps
    pairs = nq * ndb
    rate = pairs / sec
    peak_pairs = VALU_LANE_OPS_PER_S / OPS_PER_PAIR
    hb, h1, h2 = best.cpu().numpy(), d1.cpu().numpy().astype(np.int64), d2.cpu().numpy().astype(np.int64)
    # bit-exact check of a sample of queries against a numpy pass over the whole database
    ok = True
    for i in range(0, nq, 10):
        ok = ok and numpy_top2(hq[i], hdb) == (int(hb[i]), int(h1[i]), int(h2[i]))
    accepted = int(m.ratio_test(h1.astype(np.uint16), h2.astype(np.uint16))[1])
    return {"pairs_per_s": rate, "ms_per_call": round(sec * 1e3, 4), "queries": nq, "database_rows": ndb,
            "accepted_matches": accepted, "ratio": 0.75, "th": 50, "self_check": bool(ok),
            "self_check_what": "every tenth query against a numpy single pass over all rows (index, d1, d2)",
            "roofline": {"bound": "mfma", "dtype": "fp4", "achieved": round(rate * MAC_PER_PAIR * 2 / 1e12, 1),
                         "peak": round(MFMA_F4_MAC_PER_S * 2 / 1e12, 1), "unit": "TFLOP/s",
                         "frac": round(rate * MAC_PER_PAIR / MFMA_F4_MAC_PER_S, 4),
                         "frac_of_measured_mfma_rate": round(rate * MAC_PER_PAIR / MFMA_F4_MAC_MEASURED, 4),
                         "vs_valu_popcount_roofline": round(rate / peak_pairs, 4)}}


def bench_config5(orbx, torch, dist, rank, world, w, h, nfeat, block, stride, coll_dev):
    """Extraction plus SearchForInitialization-style matching (ratio 0.9, window 100, orientation check:
    src/Tracking.cc:729-730) of every consecutive frame pair of a sequence of world * block frames cut into
    contiguous blocks, one per rank.  A rank extracts its block plus the first frame of the next block (the
    halo) and matches the pairs whose first frame it owns: no collective on the data path.  Synthetic
    sequence: frame t+1 is frame t's scene shifted by 3 pixels (a wider canvas is generated once per rank
    and windows of it are the frames), so real matches exist."""
    from orbslam2_amd.sharded import pair_block
    n = world * block
    lo, hi_ext, pair_hi = pair_block(n, rank, world)
    nfr = hi_ext - lo
    # one canvas per rank wide enough for its frames; global frame t is the window at x = 3 * (t - lo) of the
    # canvas of the rank that owns t, and the halo frame is rebuilt from the next rank's canvas seed
    def canvas(r, count):
        return orbx.synth_image(0x5EC0DE00 + r, w + 3 * count, h, 0)
    own = canvas(rank, block)
    host = np.zeros((nfr, h, stride), np.uint8)
    for i in range(min(nfr, block)):
        host[i, :, :w] = own[:, 3 * i:3 * i + w]
    if nfr > block:
        host[block, :, :w] = canvas(rank + 1, block)[:, :w]
    dev = torch.from_numpy(host).cuda()
    torch.cuda.synchronize()
    npair = pair_hi - lo
    # The rank's pairs are cut into two contiguous chunks, each with an extractor and a matcher of its own on one
    # stream: a chunk extracts its frames plus the first frame of the next chunk (the same halo rule one level down:
    # one frame of 1024 is extracted twice) and matches its own pairs; the extractors take turns in one
    # orbx_pipeline and nothing crosses between the chunks.  Chunks 1 / 2 / 3 / 4 / 8: 175 / 186 / 164 / 178 / 145 k
    # frames/s (the matcher's workgroups want most of a compute unit's LDS and wait for the extraction kernels of
    # the other chunks to drain: more chunks, more waiting).  ORBX_BENCH_C5_CHAINS: comparison runs.
    nchain = int(os.environ.get("ORBX_BENCH_C5_CHAINS", "2"))
    nchain = max(1, min(nchain, npair))
    cuts = [npair * c // nchain for c in range(nchain + 1)]
    bounds = (0.0, float(w), 0.0, float(h))
    pipeline = orbx.Pipeline(torch.cuda.current_device()) if nchain > 1 else None
    chains = []
    for c in range(nchain):
        p0, p1 = cuts[c], cuts[c + 1]
        f0, f1 = (p0, p1 + 1) if npair > 0 else (0, nfr)  # frames p0 .. p1 hold pairs p0 .. p1-1
        st = torch.cuda.Stream()  # extractor and matcher share one real stream (the null stream would give each its own)
        ex = orbx.ORBextractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=f1 - f0,
                               device=torch.cuda.current_device())
        ex.set_stream(st.cuda_stream)
        if pipeline is not None:
            ex.set_pipeline(pipeline)
        m = orbx.ORBmatcher(0.9, True, device=torch.cuda.current_device())
        m.set_stream(st.cuda_stream)
        src = dev.data_ptr() + f0 * h * stride
        ex.extract_batch_device(src, f1 - f0, w, h, stride, h * stride)
        d_kp, d_desc, d_cnt, cap = ex.device_results()
        np_c = p1 - p0
        chains.append({"ex": ex, "m": m, "st": st, "src": src, "nfr": f1 - f0, "npair": np_c, "res": (d_kp, d_desc, d_cnt, cap),
                       "m12": torch.zeros((max(np_c, 1), cap), dtype=torch.int32, device="cuda"),
                       "nm": torch.zeros(max(np_c, 1), dtype=torch.int32, device="cuda"),
                       "npd": torch.zeros(max(np_c, 1), dtype=torch.int32, device="cuda")})

    def run():
        for ch in chains:
            ch["ex"].extract_batch_device(ch["src"], ch["nfr"], w, h, stride, h * stride)
            if ch["npair"] > 0:
                d_kp, d_desc, d_cnt, cap = ch["res"]
                # the chunk's frames are pairs (i, i+1), i < npair, of its own batch
                ch["m"].search_for_initialization_batch_device(d_kp, d_desc, d_cnt, cap, ch["npair"] + 1, 600, bounds,
                                                               ch["m12"].data_ptr(), ch["nm"].data_ptr(),
                                                               ch["npd"].data_ptr(), windowSize=100)

    for _ in range(2):
        run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=coll_dev)
    nmatch = sum(float(ch["nm"][:ch["npair"]].sum().item()) for ch in chains)
    ndist = sum(float(ch["npd"][:ch["npair"]].sum().item()) for ch in chains)
    tot = torch.tensor([float(npair), nmatch, ndist, float(sum(ch["nfr"] for ch in chains))], dtype=torch.float64,
                       device=coll_dev)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    sec = float(el.item()) / reps
    pairs, matches, dists, extracted = [float(v) for v in tot.tolist()]
    for ch in chains:
        ch["ex"].close()
    return {"frames": n, "frame_pairs": int(pairs), "frames_extracted_incl_halo": int(extracted), "chains_per_gpu": nchain,
            "frames_per_s": round(n / sec, 1), "candidate_pair_distances_per_s": round(dists / sec, 1),
            "mean_matches_per_pair": round(matches / max(pairs, 1), 1), "ms_per_pass": round(sec * 1e3, 3),
            "sharding": "contiguous frame blocks, 1-frame halo re-extracted by the owner of the pair, no collective"}


def bench_hamming_sharded(orbx, torch, dist, stream, rank, world, coll_dev):
    """2000 queries x 1M rows, rows sharded over the ranks, per-shard top-2 exchanged with one all-gather
    (RCCL) and merged on every rank.  With the nccl backend the whole step is the library's C path
    (orbx_match_top2_sharded_device: kernels and ncclAllGather enqueued by C code on one stream); a gloo
    rehearsal, where ranks share a GPU and RCCL cannot run, exchanges the blocks through torch."""
    from orbslam2_amd.sharded import RcclShardedMatcher, ShardedMatcher, shard_bounds
    nq, ndb = 2000, 1_000_000
    lo, hi = shard_bounds(ndb, rank, world)
    full, q, _ = config4_data(orbx, nq, ndb)
    db = torch.from_numpy(full[lo:hi].copy()).cuda()
    dq = torch.from_numpy(q).cuda()
    c_path = coll_dev == "cuda"
    if c_path:
        sm = RcclShardedMatcher(rank, world, torch.cuda.current_device(), max_queries=nq)
        best = torch.zeros(nq, dtype=torch.int32, device="cuda")
        d1 = torch.zeros(nq, dtype=torch.int16, device="cuda")
        d2 = torch.zeros(nq, dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()  # inputs complete before the handle's own stream reads them

        def run():
            sm.match_device(dq.data_ptr(), nq, db.data_ptr(), hi - lo, lo, best.data_ptr(), d1.data_ptr(), d2.data_ptr())
            return best, d1, d2

        def sync():
            sm.synchronize()
    else:
        # the matcher on this rank's GPU: its launches make that device current
        m = orbx.ORBmatcher(0.75, device=torch.cuda.current_device())
        tm = ShardedMatcher(None, rank, world, device=coll_dev, matcher=m)

        def run():
            return tm.match_device(dq, nq, db, hi - lo, lo)

        def sync():
            torch.cuda.synchronize()
    for _ in range(3):
        res = run()
    sync()
    dist.barrier()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        res = run()
    sync()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=coll_dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    sec = float(dt.item()) / reps
    res = [res[0].cpu().numpy(), res[1].cpu().numpy().view(np.uint16), res[2].cpu().numpy().view(np.uint16)]
    ok = True
    if rank == 0:
        for i in range(0, nq, nq // 20):
            ok = ok and numpy_top2(q[i], full) == (int(res[0][i]), int(res[1][i]), int(res[2][i]))
    del full
    if c_path:
        sm.close()
    rate = nq * ndb / sec
    return {"pairs_per_s": rate, "ms_per_call": round(sec * 1e3, 4), "queries": nq, "database_rows": ndb,
            "shards": world,
            "exchange": ("orbx_match_top2_sharded_device: ncclAllGather of 8 B per query and rank on the kernels' stream, "
                         "merge kernel" if c_path else
                         "torch all_gather of 8 B per query and rank through host memory (gloo), merge kernel"),
            "self_check": ok,
            "roofline": {"bound": "mfma", "dtype": "fp4", "achieved": round(rate * MAC_PER_PAIR * 2 / 1e12 / world, 1),
                         "peak": round(MFMA_F4_MAC_PER_S * 2 / 1e12, 1), "unit": "TFLOP/s per GPU",
                         "frac": round(rate / world * MAC_PER_PAIR / MFMA_F4_MAC_PER_S, 4)}}


def pcie_in_child(args, env, local_rank):
    """`bench.py --pcie-only` as a child process with the caller's original environment (no launcher variables, so
    it is a one-rank run on this rank's GPU); returns its JSON object, or an error entry."""
    import subprocess
    env = {k: v for k, v in env.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT")
           and not k.startswith("TORCHELASTIC")}
    env["LOCAL_RANK"] = str(local_rank)
    cmd = [sys.executable, os.path.abspath(__file__), "--pcie-only", "--gpus", "1", "--workload", args.workload,
           "--pcie-instances", str(args.pcie_instances), "--batch", "256", "--frames-resident", "256", "--hw-queues", "0"]
    try:
        out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240, check=True)
        line = [l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:  # the headline does not depend on it
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def pcie_inclusive(orbx, torch, host, w, h, nfeat, device, batch=64, instances=2, rounds=24):
    """Frames start in host memory and keypoints / descriptors end in host memory: what a caller of
    operator() on host images pays.  `instances` extractors, each with its own stream, take batches in turn,
    so that the upload and extraction of one batch overlap the download of another (a stereo Frame runs two
    extractors, src/Frame.cc:79-82; a frame server runs more).  Never the benchmark's `value`."""
    import ctypes as C

    from orbslam2_amd import _lib as L

    lib = L.lib()
    nfr = min(len(host), batch * 4)
    stride = host.shape[2]
    res = {"batch": batch, "instances": instances, "frame_bytes": w * h}
    exs = [orbx.ORBextractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=batch, device=device)
           for _ in range(instances)]
    cap = exs[0].max_keypoints
    kp = [torch.empty((batch, cap * 28), dtype=torch.uint8).pin_memory() for _ in range(instances)]
    de = [torch.empty((batch, cap, 32), dtype=torch.uint8).pin_memory() for _ in range(instances)]
    cnt = [np.zeros(batch, np.int32) for _ in range(instances)]
    for kind in ("pinned", "pageable"):
        if kind == "pinned":
            src = torch.from_numpy(host[:nfr]).pin_memory()
            base, fbytes = src.data_ptr(), h * stride
        else:
            src = host[:nfr].copy()
            base, fbytes = src.ctypes.data, h * stride
        ptrs = [(C.c_void_p * batch)(*[base + ((b * batch + f) % nfr) * fbytes for f in range(batch)])
                for b in range(nfr // batch)]

        def submit(i, b):
            L.check(lib.orbx_extract_batch_host(exs[i]._h, ptrs[b % len(ptrs)], batch, w, h, stride),
                    "orbx_extract_batch_host")

        def collect(i):
            L.check(lib.orbx_batch_download_all(exs[i]._h, C.c_void_p(kp[i].data_ptr()), C.c_void_p(de[i].data_ptr()),
                                                cap, L.ptr(cnt[i])), "orbx_batch_download_all")

        total_kp = 0
        for timed in (False, True):
            n = rounds if timed else instances * 2
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for it in range(n):
                i = it % instances
                if it >= instances:
                    collect(i)
                    total_kp += int(cnt[i].sum()) if timed else 0
                submit(i, it)
            for it in range(n, n + instances):
                collect(it % instances)
                total_kp += int(cnt[it % instances].sum()) if timed else 0
            dt = time.perf_counter() - t0
        fps = rounds * batch / dt
        res[kind] = {"frames_per_s": round(fps, 1), "h2d_GBps": round(fps * w * h / 1e9, 2),
                     "d2h_GBps": round(total_kp / dt * 60 / 1e9, 2), "ms_per_batch": round(dt / rounds * 1e3, 3)}
    for e in exs:
        e.close()
    return res


REFERENCE_CFLAGS = ["-O3", "-march=native"]  # what the reference builds with (CMakeLists.txt:10-11 of the reference)


def timing_build_of_oracle():
    """A second, timing-only build of the oracle with the reference's compiler flags, made on the machine that
    runs it (-march=native must see this host's CPU).  The checker build (-O2 -ffp-contract=off, every float
    operation rounded separately) stays the one results are compared with.  Returns (extract function, flags)
    or (None, reason)."""
    import ctypes as C
    import shutil
    import subprocess
    import tempfile

    from oracle import pyoracle as orc

    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        return None, "no C compiler on this machine"
    out = os.path.join(tempfile.mkdtemp(prefix="orbx_cpu_baseline_"), "liborb_oracle_native.so")
    cmd = [cc] + REFERENCE_CFLAGS + ["-fPIC", "-std=c11", "-shared", "-o", out,
                                     os.path.join(ROOT, "oracle", "orb_oracle.c"), "-lm"]
    try:
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        h = C.CDLL(out)
    except (subprocess.CalledProcessError, OSError) as exc:
        return None, repr(exc)
    h.orc_extract.argtypes = [C.POINTER(orc.Params), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                              C.c_int, C.POINTER(C.c_int), C.c_void_p]

    def extract(img, nfeat):
        p = orc.Params(nfeat, 1.2, 8, 20, 7, 0)
        cap = nfeat + 96
        kps = np.zeros(cap, orc.KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        rc = h.orc_extract(C.byref(p), img.ctypes.data, img.shape[1], img.shape[0], img.strides[0], kps.ctypes.data,
                           desc.ctypes.data, cap, C.byref(n), None)
        assert rc == 0 and 0 < n.value <= cap
    return extract, " ".join(REFERENCE_CFLAGS)


def cpu_baseline(host_frames, w, nfeat, budget_s):
    """The oracle (scalar C restatement of the reference extractor) on one host core, same frames, built with
    the reference's own compiler flags (-O3 -march=native) for the timing; the checker build's rate beside it;
    then the timing build frame-parallel on the host cores this process may use (the reference runs one thread
    per image: src/Frame.cc:79-82)."""
    import concurrent.futures

    from oracle import pyoracle as orc

    orc.lib()
    imgs = [np.ascontiguousarray(f[:, :w]) for f in host_frames[:64]]
    native, flags = timing_build_of_oracle()

    def one_checker(i):
        orc.extract(imgs[i % len(imgs)], nfeat, 1.2, 8, 20, 7)  # ctypes releases the GIL during the call

    def one_native(i):
        native(imgs[i % len(imgs)], nfeat)

    def timed(fn, seconds):
        n = 0
        t0 = time.perf_counter()
        while True:
            fn(n)
            n += 1
            dt = time.perf_counter() - t0
            if dt >= seconds or n >= 2000:
                return n, dt

    caveat = ("scalar C: no SIMD FAST / resize / blur as OpenCV has, so still pessimistic for the reference by an "
              "unmeasured factor")
    n, dt = timed(one_checker, budget_s * (0.25 if native else 0.6))
    checker = {"value": round(n / dt, 2), "flags": "-O2 -ffp-contract=off -fno-fast-math",
               "sample": "%d frames in %.1f s, one thread" % (n, dt)}
    if native is None:
        res = {"value": checker["value"], "unit": "frames/s", "cores": 1, "kind": "port", "flags": checker["flags"],
               "sample": "%d KITTI-shape synthetic frames in %.1f s, one thread; %s" % (n, dt, caveat),
               "timing_build": "unavailable: %s" % flags}
        one = one_checker
    else:
        n, dt = timed(one_native, budget_s * 0.35)
        res = {"value": round(n / dt, 2), "unit": "frames/s", "cores": 1, "kind": "port", "flags": flags,
               "sample": "%d KITTI-shape synthetic frames in %.1f s, one thread, the oracle compiled on this host "
                         "with the reference's flags; %s" % (n, dt, caveat),
               "checker_build": checker}
        one = one_native
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    if threads > 1:
        want = max(threads, int(res["value"] * threads * budget_s * 0.4))
        t0 = time.perf_counter()
        with concurrent.futures.ThreadPoolExecutor(threads) as pool:
            list(pool.map(one, range(want)))
        dt = time.perf_counter() - t0
        res["frame_parallel"] = {"value": round(want / dt, 2), "unit": "frames/s", "cores": threads,
                                 "sample": "%d frames in %.1f s, one frame per thread" % (want, dt)}
    return res


if __name__ == "__main__":
    main()
