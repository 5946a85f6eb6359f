"""HBM traffic of bench workloads by rocprofv3 counter passes, written to profiles/pmc_traffic.json
together with the blob hashes of the kernel sources they ran on.

    python tools/traffic_stamp.py --profile profiles/r03_traffic.txt whisper128 clap64 ...

FETCH_SIZE and WRITE_SIZE are collected in separate runs (they do not fit one pass), summed over
the library's kernels of one step; FETCH_SIZE is doubled (on gfx950 it counts 64 of every 128
bytes of a wide streaming read).  Run on the GPU box from the repository root.
"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mlx_audio_amd import provenance  # noqa: E402


def one_pass(workload, counter, out):
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", out, "--",
           "python3", os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-secondary"]
    with open(out + ".log", "w") as log:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, timeout=170,
                       check=True)
    line = None
    for text in open(out + ".log"):
        if text.startswith("{") and '"metric"' in text:
            line = json.loads(text)
    per_kernel = defaultdict(lambda: [0, 0.0])
    for path in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if "mia::" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                k = row["Kernel_Name"].split("(")[0][:70]
                per_kernel[k][0] += 1
                per_kernel[k][1] += float(row["Counter_Value"])
    # calls of the step in this run: the untimed clock warm-up, the warm-up steps and the timed ones;
    # a kernel's dispatches beyond a whole number per call belong to the workload's set-up
    calls = line["config"]["clock_warmup_steps"] + max(line["warmup"], 1) + line["steps"]
    detail = {k: {"kib_per_dispatch": v / n, "dispatches_per_step": n // calls}
              for k, (n, v) in sorted(per_kernel.items()) if n // calls > 0}
    total = sum(d["kib_per_dispatch"] * d["dispatches_per_step"] for d in detail.values())
    return total, detail, line


def main():
    args = sys.argv[1:]
    profile = None
    if args and args[0] == "--profile":
        profile, args = args[1], args[2:]
    path = os.path.join(ROOT, provenance.TRAFFIC_FILE)
    table = json.load(open(path))
    report = []
    for w in args:
        out = os.path.join(ROOT, "gpurun_out", "traffic_" + w)
        os.makedirs(out, exist_ok=True)
        fetch, fd, line = one_pass(w, "FETCH_SIZE", os.path.join(out, "FETCH_SIZE"))
        write, wd, _ = one_pass(w, "WRITE_SIZE", os.path.join(out, "WRITE_SIZE"))
        clips = line["config"]["clips_per_gpu"]
        abytes = line["roofline"]["algorithmic_bytes_per_step"]
        hbm = int((2 * fetch + write) * 1024)
        table[f"{w}:{clips}"] = {
            "kernel": ", ".join(sorted(fd)),
            "fetch_size_kib": round(fetch),
            "write_size_kib": round(write),
            "hbm_bytes_per_step": hbm,
            "algorithmic_bytes_per_step": abytes,
            "kernel_sources": provenance.source_hashes(w),
            "profile": profile,
        }
        report.append(f"{w}:{clips}  fetch {fetch:.0f} KiB (x 2 on gfx950)  write {write:.0f} KiB  -> "
                      f"{hbm} B per step = {hbm / abytes:.3f} x the algorithmic {abytes} B")
        for c, d in (("FETCH_SIZE", fd), ("WRITE_SIZE", wd)):
            for k, v in d.items():
                report.append(f"    {c} {k}: {v['kib_per_dispatch']:.0f} KiB per dispatch, "
                              f"{v['dispatches_per_step']:g} dispatches per step")
        print(report[-1 - 0], flush=True)
    json.dump(table, open(path, "w"), indent=1)
    text = "\n".join(report) + "\n"
    print(text)
    if profile:
        with open(os.path.join(ROOT, profile), "a") as f:
            f.write(text)


if __name__ == "__main__":
    main()
