"""profiles/r0N_pmc_meta.json (N = argv[1], default 4) from the committed PMC passes (scripts/gpu_r0N_profiles.sh): for every bench leg, the HBM bytes per
full-size launch of each kernel bench.py can report as dominant, its average duration in the kernel trace of the same command, and
its launch count per step -- bench.py quotes `roofline.traffic` (and prices `roofline.frac` with it) only for a run with the same
walkers AND the same launch count of that kernel per step.
FETCH_SIZE / WRITE_SIZE are KiB; FETCH is DOUBLED (gfx950: MI355X_MICROARCH.md, HBM).  The profiled command runs the path four
times (calibration with 1 walker, warm-up and timed step at full size, rank diagnostics with 16 walkers): the full-size launches
are identified by their grid in the kernel trace (the two largest launch groups of the kernel) and carry all but ~0.1 % of the bytes."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("chol_pivot_kernel", "rows_qr_kernel", "tgemm_chain_kernel", "tgemm_chain3_kernel", "tgemm_direct_kernel", "tgemm_skinny_f64_kernel", "gram_cols_i8_kernel", "gram_cols_f64_kernel", "gram_cols_lds_kernel", "gram_rows_f64_kernel", "chol_blocked_kernel",
           "gram_chol_wave_kernel", "jacobi_rows_grp_kernel", "jacobi_rows_tiny4_kernel", "colgram_dense_kernel", "mid_gram_chol_kernel",
           "ortho_rows_kernel", "mgemm_dense_kernel", "tgemm_kernel")
RND = "r%02d" % (int(sys.argv[1]) if len(sys.argv) > 1 else 4)


def sq_shares(path):
    """kernel -> {wait_any, wait_inst_any, active_valu: shares of SQ_WAVE_CYCLES; valu_per_mfma; mfma_busy_over_cu_busy} from the SQ pass"""
    raw = {}
    if not os.path.exists(path):
        return {}
    for line in open(path):
        f = line.split()
        if len(f) < 5 or not f[-4].startswith("SQ_"):
            continue
        raw.setdefault(" ".join(f[:-4]), {})[f[-4]] = float(f[-2])
    out = {}
    for k, c in raw.items():
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0:
            continue
        ent = {"wait_any": c.get("SQ_WAIT_ANY", 0.0) / wc, "wait_inst_any": c.get("SQ_WAIT_INST_ANY", 0.0) / wc,
               "active_inst_valu": c.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, "wave_cycles": wc}
        if c.get("SQ_INSTS_MFMA", 0.0) > 0:
            ent["valu_per_mfma"] = c.get("SQ_INSTS_VALU", 0.0) / c["SQ_INSTS_MFMA"]
            ent["insts_mfma"] = c["SQ_INSTS_MFMA"]
        if c.get("SQ_BUSY_CU_CYCLES", 0.0) > 0:
            ent["mfma_busy_over_cu_busy"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / c["SQ_BUSY_CU_CYCLES"]
        out[k.replace("void ", "").replace("pepsgpu::", "")] = ent
    return out
LEGS = {"c4_f32_noise0.1": "c4_f32_noise0.1_nw49152", "c4_f32_noise1": "c4_f32_noise1_nw8192",
        "c4_f32_real": "c4_f32_real_nw12288" if RND >= "r06" else "c4_f32_real_nw8192"}


def totals(path):
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        f = line.split()
        if len(f) < 5 or f[-4] not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        out[" ".join(f[:-4])] = (int(f[-3]), float(f[-2]))
    return out


def trace(path):
    """kernel -> list of (ms, calls, us_per_call, grid) from the by-grid summary"""
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        m = re.match(r"\s*([\d.]+) ms\s+(\d+) calls\s+([\d.]+) us/call\s+\('(.*?)', '(\d+)'", line)
        if m:
            out.setdefault(m.group(4), []).append((float(m.group(1)), int(m.group(2)), float(m.group(3)), int(m.group(5))))
    return out


def main():
    meta = {}
    for leg, tag in LEGS.items():
        fe = totals(os.path.join(ROOT, "profiles", RND + "_pmc_FETCH_SIZE_%s.txt" % tag))
        wr = totals(os.path.join(ROOT, "profiles", RND + "_pmc_WRITE_SIZE_%s.txt" % tag))
        tr = trace(os.path.join(ROOT, "profiles", RND + "_kernel_trace_by_grid_%s.txt" % tag))
        cfg = os.path.join(ROOT, "profiles", RND + "_bench_profiled_config_%s.json" % tag)
        if not (fe and wr and tr and os.path.exists(cfg)):
            continue
        bench = json.load(open(cfg))
        ent = {"walkers": bench["config"]["walkers_per_gpu"], "source": "profiles/%s_pmc_{FETCH,WRITE}_SIZE_%s.txt + %s_kernel_trace_by_grid_%s.txt" % (RND, tag, RND, tag),
               "kernels": {}}
        for k in KERNELS:
            kf = sum(v[1] for n, v in fe.items() if k in n)
            kw = sum(v[1] for n, v in wr.items() if k in n)
            # (the by-grid summary keeps the last 60 characters of a kernel name: the longer template lists of round 5 cut the head
            # of "void pepsgpu::tgemm_chain_kernel<...>" down to "gemm_chain_kernel<...>" -- a left-truncated name still matches)
            def hit(n):
                return k in n or any(n.startswith(k[j:] + "<") or n.startswith(k[j:] + " ") for j in range(1, 8))
            groups = [g for n, gs in tr.items() if hit(n) for g in gs]
            if not groups:
                continue
            gmax = max(g[3] for g in groups)
            full = [g for g in groups if g[3] == gmax]           # the full-size launches (largest grid)
            calls = sum(g[1] for g in full)
            ms = sum(g[0] for g in full)
            if calls == 0:
                continue
            # profiled command = warm-up step + timed step at full size: launches per step = calls / 2
            ent["kernels"][k] = {"hbm_bytes_per_launch": (2.0 * kf + kw) * 1024.0 / calls, "launches_per_step": calls / 2.0,
                                 "avg_us": 1e3 * ms / calls, "fetch_KiB": kf, "write_KiB": kw}
        ent["sq"] = sq_shares(os.path.join(ROOT, "profiles", RND + "_pmc_SQ_%s.txt" % tag))
        meta[leg] = ent
    json.dump(meta, open(os.path.join(ROOT, "profiles", RND + "_pmc_meta.json"), "w"), indent=1)
    print(json.dumps(meta, indent=1)[:3000])


if __name__ == "__main__":
    main()
