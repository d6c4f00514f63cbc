#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv output into profiles/pmc_traffic.json.

    tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <bench stdout of the run> <out.json>

Per bench stage, HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / launches: FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section), hence the factor 2 (an upper bound for
narrow accesses).  A 'launch' is what bench.py's stage timer counts: one row-FFT pass, one
scatter / gather pass, one pad or crop pass.
"""

import collections
import csv
import json
import sys

STAGE_OF = [
    ("k_grid", "grid"), ("k_degrid", "degrid"), ("k_fused_fft_crop", "fft_crop"), ("k_fused_pad_fft", "pad_fft"),
    ("k_rowfft", "fft_rows"), ("fft_", "fft_rows"), ("transpose_", "fft_rows"),
    ("k_pad_screen", "pad"), ("k_b2a", "pad"), ("k_crop_screen", "crop"), ("k_a2b", "crop"),
]


def collect(path, counter):
    tot = collections.Counter()
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for key, stage in STAGE_OF:
            if key in r["Kernel_Name"]:
                tot[stage] += float(r["Counter_Value"])
                disp[stage].add(r["Dispatch_Id"])
                break
    return tot, {s: len(v) for s, v in disp.items()}


def main():
    fetch, _ = collect(sys.argv[1], "FETCH_SIZE")
    write, _ = collect(sys.argv[2], "WRITE_SIZE")
    bench = None
    for line in open(sys.argv[3]):
        if line.startswith('{"metric"'):
            bench = json.loads(line)
    applies = bench["steps"] + bench["warmup"]  # every apply of the profiled process, warm-up included
    per_step = bench["roofline"]["stage_launches_per_step"]  # launches as bench.py's stage timers count them
    out = {}
    for s in fetch:
        out[s] = (2.0 * fetch[s] + write.get(s, 0.0)) * 1024.0 / (per_step[s] * applies)
    out["_kernels"] = bench.get("roofline", {}).get("stage_kernels", {})  # the kernel each stage timer bracketed in that run
    out["_note"] = ("HBM bytes per stage launch = (2*FETCH_SIZE + WRITE_SIZE) KiB / (stage launches per apply x applies), "
                    "rocprofv3 --pmc in two separate passes; gfx950 FETCH_SIZE x2 correction applied (upper bound)")
    import subprocess, time
    out["_collected_unix"] = int(time.time())  # bench.py picks the newest profile by this (mtimes do not survive a checkout)
    try:
        out["_collected_at_commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:
        out["_collected_at_commit"] = None
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
