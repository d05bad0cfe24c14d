# kernel-by-kernel timeline of one full-rank step: which launches are the expensive ones (name, grid, duration)
cd $GRAFT_REPO_ROOT; O=gpurun_out/frseq; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -o kt -- python3 bench.py --noise ${NOISE:-1.0} --walkers ${NW:-4096} --steps 1 --warmup 1 --no-cpu-baseline --no-route-check --no-full-rank --no-energy-check > $O/kt.log 2>&1
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/frseq/kt_kernel_trace.csv')))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(n):
    n = n.replace('void pepsgpu::', '').replace('pepsgpu::', '')
    return n[:44]
# histogram of durations per kernel for full-size grids
h = collections.defaultdict(list)
for r in rows:
    h[nm(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(h.items(), key=lambda kv: -sum(kv[1]))[:12]:
    v2 = sorted(v)
    print("%-46s n=%4d tot=%8.1f ms  p10=%7.1f p50=%7.1f p90=%7.1f max=%7.1f us" % (k, len(v), sum(v) / 1e3, v2[len(v2) // 10], v2[len(v2) // 2], v2[(9 * len(v2)) // 10], v2[-1]))
# the timeline of the second half of the run (timed step), launches above 150 us
t = [(nm(r["Kernel_Name"]), r["Grid_Size_X"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
half = t[len(t) // 2:]
out = open('gpurun_out/frseq/timeline.txt', 'w')
for k, g, d in half:
    out.write("%-46s %9s %9.1f\n" % (k, g, d))
PY
rm -f $O/kt_kernel_trace.csv
