"""profiles/r02_pmc_meta.json from the committed PMC passes (scripts/gpu_r02_profiles.sh): HBM bytes per full-size launch of
every kernel category bench.py can report as dominant, and the launch count per step of the profiled run -- bench.py quotes
`roofline.traffic` only for a run with the same workload AND the same launch count in that category.
FETCH_SIZE / WRITE_SIZE are KiB; FETCH is doubled (gfx950: MI355X_MICROARCH.md, HBM).  The profiled command runs the path four
times (calibration with 1 walker, warm-up and timed step at full size, rank diagnostics with 16 walkers): half of the launches
are full-size and carry all but ~0.1 % of the bytes."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CATS = {"contract": ("tgemm_direct_kernel", "tgemm_chain_kernel"), "gram_f64": ("gram_cols_f64_kernel",),
        "cholesky": ("gram_chol_lowrank_kernel", "gram_chol_wave_kernel", "chol_upper_kernel", "chol_lowrank_kernel", "colgram_chol_kernel", "colgram_dense_kernel"),
        "jacobi": ("jacobi_rows_regx_kernel", "jacobi_rows_grp_kernel", "jacobi_rows_reg256_kernel"),
        "jacobi_edge": ("jacobi_rows_tiny2_kernel", "jacobi_rows_tiny4_kernel", "jacobi_rows_tiny_kernel", "jacobi_rows_small_kernel")}


def totals(path):
    out = {}
    for line in open(path):
        f = line.split()
        if len(f) < 5 or f[-4] not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        name = " ".join(f[:-4])
        out[name] = (int(f[-3]), float(f[-2]))
    return out


def main(tag, workload, dtype, walkers, noise):
    fe = totals(os.path.join(ROOT, "profiles", "r02_pmc_FETCH_SIZE_%s.txt" % tag))
    wr = totals(os.path.join(ROOT, "profiles", "r02_pmc_WRITE_SIZE_%s.txt" % tag))
    bench = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_profiled_config_%s.json" % tag)))
    meta = {"workload": workload, "dtype": dtype, "walkers": walkers, "noise": noise, "source": "profiles/r02_pmc_{FETCH,WRITE}_SIZE_%s.txt" % tag,
            "categories": {}}
    for cat, keys in CATS.items():
        lf = sum(v[0] for k, v in fe.items() if any(x in k for x in keys))
        kf = sum(v[1] for k, v in fe.items() if any(x in k for x in keys))
        kw = sum(v[1] for k, v in wr.items() if any(x in k for x in keys))
        if lf == 0 or cat not in bench.get("kernel_ms", {}):
            continue
        full = lf / 2.0
        meta["categories"][cat] = {"hbm_bytes_per_launch": (2.0 * kf + kw) * 1024.0 / full, "kernel_launches_full_size": full,
                                   "fetch_KiB": kf, "write_KiB": kw}
    # launches per step as bench.py counts them (one profiling bracket per launch site)
    steps = bench["steps"]
    rl = bench["roofline"]
    meta["categories"].setdefault(rl["kernel"], {})["launches_per_step"] = rl["launches"] / steps
    return meta


if __name__ == "__main__":
    m = main("c4_f32_nw32768", "C4", "f32", 32768, 0.1)
    json.dump(m, open(os.path.join(ROOT, "profiles", "r02_pmc_meta.json"), "w"), indent=1)
    print(json.dumps(m, indent=1))
