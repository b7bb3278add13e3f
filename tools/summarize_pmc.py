#!/usr/bin/env python3
"""Turns the counter passes written by tools/collect_profiles.sh into profiles/traffic.json and a
text table.  Usage: python tools/summarize_pmc.py gpurun_out/r01 profiles/r01

HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB, and on gfx950 FETCH_SIZE
reports half of the bytes read, which the calibration kernels of tools/ubench_fetch.hip (1 GiB
streamed with 4-byte and with 16-byte loads per lane) confirm for the access widths used here.
"""
import collections
import csv
import glob
import json
import os
import sys


def load(d):
    files = glob.glob(os.path.join(d, "vm", "*counter_collection.csv")) + glob.glob(
        os.path.join(d, "*", "*counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    files.sort(key=os.path.getmtime)  # a directory merged from several runs: the newest pass counts
    for r in csv.DictReader(open(files[-1])):
        # "void k_octree<256, 14>(...)" and "k_fast(...)" alike: the bare kernel name
        name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").strip()
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def mean(x):
    return sum(x) / len(x)


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    bench = json.loads(open(os.path.join(src, "bench_exclusive.json")).read().strip().splitlines()[-1])
    frames = bench["frames_per_launch"]
    fetch, write, sq = load(os.path.join(src, "pmc_fetch")), load(os.path.join(src, "pmc_write")), load(
        os.path.join(src, "pmc_sq"))
    cal_f, cal_w = load(os.path.join(src, "cal_fetch")), load(os.path.join(src, "cal_write"))
    lines = ["counter calibration (1 GiB = 1048576 KiB streamed per kernel):"]
    for k in ("read_b32", "read_b128", "write_b32", "write_b128"):
        lines.append("  %-10s FETCH_SIZE %10.1f KiB   WRITE_SIZE %10.1f KiB" % (
            k, mean(cal_f[k]["FETCH_SIZE"]), mean(cal_w[k]["WRITE_SIZE"])))
    lines.append("")
    lines.append("per launch of %d frames (%s), one extractor instance, stages serialised:" % (
        frames, bench["config"]["workload"]))
    lines.append("%-11s %8s %12s %12s %12s %12s %12s %10s" % (
        "kernel", "launches", "FETCH KiB", "WRITE KiB", "HBM MB", "VALU instr", "SALU instr", "LDS instr"))
    kernels = {}
    for k in ("k_resize", "k_fast", "k_octree", "k_blur", "k_describe"):
        # launches per batch: one resize per level above 0, two quadtree launches (large and small levels), one else
        per_batch = max(1, round(len(sq[k]["SQ_INSTS_VALU"]) / len(sq["k_fast"]["SQ_INSTS_VALU"])))
        f = mean(fetch[k]["FETCH_SIZE"]) * per_batch
        w = mean(write[k]["WRITE_SIZE"]) * per_batch
        hbm = (2 * f + w) * 1024
        valu = mean(sq[k]["SQ_INSTS_VALU"]) * per_batch
        salu = mean(sq[k]["SQ_INSTS_SALU"]) * per_batch
        lds = mean(sq[k]["SQ_INSTS_LDS"]) * per_batch
        lines.append("%-11s %8d %12.1f %12.1f %12.2f %12.0f %12.0f %10.0f" % (
            k, per_batch, f, w, hbm / 1e6, valu, salu, lds))
        kernels[k] = {"hbm_bytes_per_frame": hbm / frames, "fetch_kib_per_launch": f, "write_kib_per_launch": w,
                      "valu_wave_instr_per_frame": valu / frames, "salu_wave_instr_per_frame": salu / frames,
                      "lds_wave_instr_per_frame": lds / frames}
    l1dir = os.path.join(src, "pmc_l1")
    if os.path.isdir(l1dir):
        l1 = load(l1dir)
        lines.append("")
        lines.append("vector memory path per launch (an average over the launches of a kernel; TA_TA_BUSY summed over the 256 compute units, GRBM_GUI_ACTIVE over the 8 XCDs):")
        lines.append("%-11s %14s %16s %16s %16s %10s" % (
            "kernel", "TA busy cyc", "L1 accesses", "L1->L2 reads", "pending stall", "TA busy"))
        for k in ("k_resize", "k_fast", "k_octree", "k_blur", "k_describe"):
            c = {n: mean(v) for n, v in l1[k].items()}
            active = c["GRBM_GUI_ACTIVE"] / 8.0
            lines.append("%-11s %14.0f %16.0f %16.0f %16.0f %9.0f%%" % (
                k, c["TA_TA_BUSY_sum"], c["TCP_TOTAL_CACHE_ACCESSES_sum"], c["TCP_TCC_READ_REQ_sum"],
                c["TCP_PENDING_STALL_CYCLES_sum"], 100.0 * c["TA_TA_BUSY_sum"] / 256.0 / active))
            kernels[k]["ta_busy_frac"] = c["TA_TA_BUSY_sum"] / 256.0 / active
            kernels[k]["l1_to_l2_read_requests_per_frame"] = c["TCP_TCC_READ_REQ_sum"] / frames
    ldsdir = os.path.join(src, "pmc_lds")
    if os.path.isdir(ldsdir):
        ld = load(ldsdir)
        lines.append("")
        lines.append("LDS per launch (an average over the launches of a kernel; SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT: cycles summed over the 256 compute units; "
                     "SQ_BUSY_CYCLES over the 32 shader engines):")
        lines.append("%-11s %14s %16s %12s %18s" % ("kernel", "LDS instr", "LDS active cyc", "conflict", "LDS busy of time"))
        for k in ("k_fast", "k_octree", "k_blur", "k_describe"):
            c = {n: mean(v) for n, v in ld[k].items()}
            busy = c["SQ_BUSY_CYCLES"] / 32.0
            lines.append("%-11s %14.0f %16.0f %11.0f%% %17.0f%%" % (
                k, c["SQ_INSTS_LDS"], c["SQ_LDS_IDX_ACTIVE"],
                100.0 * c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0),
                100.0 * c["SQ_LDS_IDX_ACTIVE"] / 256.0 / busy))
            kernels[k]["lds_bank_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0)
            kernels[k]["lds_busy_frac"] = c["SQ_LDS_IDX_ACTIVE"] / 256.0 / busy
    out = {"workload": bench["config"]["workload"], "frames_per_launch_measured": frames,
           "source": "%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_* passes of "
                     "'bench.py --exclusive', HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB" % dst,
           "kernels": kernels}
    with open(os.path.join(os.path.dirname(dst.rstrip("/")) or ".", "traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    with open(os.path.join(dst, "pmc_summary.txt"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
