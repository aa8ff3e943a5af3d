"""One place that says which family a kernel symbol belongs to (tools/measure_round.sh, tools/pmc_traffic.py, tools/prof_summary.py).

VERDICT r5: two pool kernels renamed in round 5 (avg2_rows_kernel, avgmax_rows2_kernel) fell through a `"pool" in name` test into
"other", so kernel_trace.json under-reported pool time (0.16 vs 0.64 ms) and the traffic note under-reported pool bytes."""

POOL = ("pool", "avg2_rows", "avgmax_rows")
CONV = ("csp_fused", "csp_tile")


def family(name: str) -> str:
    n = name.replace("void ", "").replace("cc::", "")
    if any(t in n for t in POOL):
        return "pool"
    if "stem_fused" in n:
        return "stem"
    if any(t in n for t in CONV) or ("conv" in n and "kernel" in n):
        return "conv"
    return "other"
