"""GPU-box tool: HBM traffic of the headline configuration from rocprofv3 PMC passes -> profiles/pmc_traffic.json.

    cd /tmp && export TMPDIR=/tmp && python $GRAFT_REPO_ROOT/tools/pmc_traffic.py [tag]

Two separate counter passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains besides the kernel trace, as
the GPU pool requires), each over `tools/dev/plan_passes.py N` = N replays of the bench plan (YOLOv9-C, B=64, 640x640, storage mode $CLEARCAM_BENCH_DTYPE, default f16h).
Per MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE count L2 memory-side requests (Infinity-Cache hits included),
rocprofv3 reports them in KiB, and on gfx950 FETCH_SIZE reports HALF the bytes of wide coalesced (16 B per lane) reads —
every conv / pool kernel here loads that way (global_load_lds_dwordx4 / dwordx4), so their read bytes are 2 x FETCH_SIZE;
WRITE_SIZE is taken as reported (uncalibrated, as the guide says).  The record carries the digest of clearcam_amd/csrc so
bench.py refuses to quote it for other kernels.  The per-kernel table goes to profiles/<tag>_pmc_traffic.txt.
"""
import csv
import glob
import json
import os
import subprocess
import sys
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PASSES = 4                                  # replays of the plan per counter pass (after 1 warm-up replay that is also counted)


def run_pass(counter: str, outdir: str):
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", outdir, "-o", "p", "--",
           sys.executable, os.path.join(ROOT, "tools", "dev", "plan_passes.py"), str(PASSES)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise SystemExit(f"rocprofv3 {counter} pass failed:\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")
    files = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)
    agg, calls = defaultdict(float), defaultdict(int)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                k = row["Kernel_Name"].replace("void ", "").replace("cc::", "")
                agg[k] += float(row["Counter_Value"]); calls[k] += 1
    return agg, calls


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    from bench import kernel_source_digest
    base = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"pmc_{int(time.time())}")
    fetch, fcalls = run_pass("FETCH_SIZE", base + "_f")
    write, wcalls = run_pass("WRITE_SIZE", base + "_w")
    replays = PASSES + 1
    # every conv kernel family (conv_mfma / conv_phase / conv_persist / conv_direct / 3x3 specialisations / fused RepNCSP); rocprofv3
    # leaves the _Float16 instantiations MANGLED (_ZN2cc19conv_persist_kernelIDF16_...), so match inside the name
    from tools.kernel_family import family
    lines, conv_r, conv_w, pool_r, pool_w = [], 0.0, 0.0, 0.0, 0.0
    for k in sorted(set(fetch) | set(write)):
        r_b, w_b = 2.0 * fetch.get(k, 0.0) * 1024 / replays, write.get(k, 0.0) * 1024 / replays
        if "stem_fused" in k:
            r_b /= 2.0                                                         # dword loads of the uint8 frames: FETCH_SIZE taken as reported
        lines.append(f"{k[:100]:100} launches/step {fcalls.get(k, 0) / replays:7.1f}  read {r_b / 1e9:8.3f} GB  write {w_b / 1e9:8.3f} GB")
        if family(k) == "conv":
            conv_r += r_b; conv_w += w_b
        elif family(k) == "pool":
            pool_r += r_b; pool_w += w_b
    dtype = os.environ.get("CLEARCAM_BENCH_DTYPE", "f16h")
    rec = {"kernel_source_digest": kernel_source_digest(), "tag": tag, "dtype": dtype, "config": f"YOLOv9-C {dtype} B=64 640x640", "plan_replays_per_pass": replays,
           "conv_read_bytes_per_step": conv_r, "conv_write_bytes_per_step": conv_w, "conv_bytes_per_step": conv_r + conv_w,
           "pool_bytes_per_step": pool_r + pool_w,
           "note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this round (profiles/{tag}_pmc_traffic.txt): conv kernels read "
                   f"{conv_r / 1e9:.2f} GB (2 x FETCH_SIZE, the gfx950 correction for 16-byte-per-lane reads) + write {conv_w / 1e9:.2f} GB per step; "
                   f"pools {(pool_r + pool_w) / 1e9:.2f} GB; Infinity-Cache hits are counted as traffic"}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(rec, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.txt"), "w") as f:
        f.write(f"# {rec['note']}\n# per kernel, per step (one replay of the plan); digest {rec['kernel_source_digest']}\n" + "\n".join(lines) + "\n")
    # the GPU box only returns gpurun_out/: leave copies there for the caller to move into profiles/
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    for name in ("pmc_traffic.json", f"{tag}_pmc_traffic.txt"):
        subprocess.run(["cp", os.path.join(ROOT, "profiles", name), os.path.join(out, name)])
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
