"""profiles/r06_mode_meta.json: for the secondary modes (f64 real state, complex real state, C5 f64 / f32) the dominant kernels of the
committed rocprofv3 runs (scripts/gpu_r06_call28.sh: --kernel-trace --stats, then FETCH_SIZE / WRITE_SIZE / SQ passes of the same probe
command) -- share of the kernel time, average launch, HBM bytes per launch ((2 FETCH + WRITE) KiB, FETCH doubled for gfx950), TB/s against
the 8 TB/s peak, SQ wait / VALU shares.  bench.py attaches the entry of a mode to its `other_modes` / `real_rank` object as `profile`."""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = "r06"
MODES = {"f64_real": "c4_f64_real_nw2048", "c128_real": "c4_c128_real_nw512", "c5_f64": "c5_f64_nw4096", "c5_f32": "c5_f32_nw4096"}


def pmc(path):
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        f = line.split()
        if len(f) < 5 or not (f[-4].startswith("SQ_") or f[-4] in ("FETCH_SIZE", "WRITE_SIZE")):
            continue
        name = " ".join(f[:-4]).replace("void ", "").replace("pepsgpu::", "")
        out.setdefault(name, {})[f[-4]] = (int(f[-3]), float(f[-2]))
    return out


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("pepsgpu::", "")[:70]


def same(a, b):
    m = min(len(a), len(b), 60)
    return a[:m] == b[:m]


meta = {}
for mode, tag in MODES.items():
    ks = os.path.join(ROOT, "profiles", "%s_kernel_stats_%s.csv" % (RND, tag))
    if not os.path.exists(ks):
        continue
    rows = list(csv.DictReader(open(ks)))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    fe, wr, sq = (pmc(os.path.join(ROOT, "profiles", "%s_pmc_%s_%s.txt" % (RND, g, tag))) for g in ("FETCH_SIZE", "WRITE_SIZE", "SQ"))
    ent = {"source": "profiles/%s_{kernel_stats,kernel_trace_by_grid,pmc_FETCH_SIZE,pmc_WRITE_SIZE,pmc_SQ}_%s.*" % (RND, tag), "kernel_ms_total": tot / 1e6, "kernels": []}
    try:
        ent["probe"] = json.load(open(os.path.join(ROOT, "profiles", "%s_probe_%s.json" % (RND, tag))))
    except Exception:
        pass
    for r in rows[:5]:
        n = short(r["Name"])
        k = {"kernel": n, "share": int(r["TotalDurationNs"]) / tot, "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
        f = next((v for kk, v in fe.items() if same(kk, n)), None)
        w = next((v for kk, v in wr.items() if same(kk, n)), None)
        if f and w and f["FETCH_SIZE"][0] > 0:
            b = (2.0 * f["FETCH_SIZE"][1] + w["WRITE_SIZE"][1]) * 1024.0
            k["hbm_bytes_per_launch_avg"] = b / f["FETCH_SIZE"][0]
            k["hbm_TB_per_s"] = b / (int(r["TotalDurationNs"]) * 1e-9) / 1e12
            k["frac_of_hbm_peak"] = k["hbm_TB_per_s"] / 8.0
        s = next((v for kk, v in sq.items() if same(kk, n)), None)
        if s and s.get("SQ_WAVE_CYCLES", (0, 0))[1] > 0:
            wc = s["SQ_WAVE_CYCLES"][1]
            k["sq"] = {"wait_any": s.get("SQ_WAIT_ANY", (0, 0))[1] / wc, "wait_inst_any": s.get("SQ_WAIT_INST_ANY", (0, 0))[1] / wc,
                       "active_inst_valu": s.get("SQ_ACTIVE_INST_VALU", (0, 0))[1] / wc,
                       "valu_per_mfma": (s["SQ_INSTS_VALU"][1] / s["SQ_INSTS_MFMA"][1]) if s.get("SQ_INSTS_MFMA", (0, 0))[1] > 0 else None}
        ent["kernels"].append(k)
    meta[mode] = ent
json.dump(meta, open(os.path.join(ROOT, "profiles", RND + "_mode_meta.json"), "w"), indent=1)
for m, e in meta.items():
    print(m, round(e["kernel_ms_total"]), "ms:", [(k["kernel"][:40], round(k["share"], 2), round(k.get("hbm_TB_per_s", 0), 2)) for k in e["kernels"][:3]])
