"""Where the device idles: reads a rocprofv3 --kernel-trace CSV (kernel_trace.csv), sorts the dispatches by start time and reports the
busy time, the span, and the idle gaps above a threshold grouped by the kernel that ran before and the one that ran after the gap.
python tools/gap_probe.py <kernel_trace.csv> [--min-us 30] [--from-kernel NAME]  (only the part of the trace from the first NAME on)"""
import argparse, csv, collections
ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--min-us", type=float, default=30.0)
ap.add_argument("--top", type=int, default=15)
ap.add_argument("--forwards", action="store_true", help="encoder forwards only: the idle time between a k_pool and the next k_embed_ln*, per forward, over the trace")
a = ap.parse_args()
rows = []
with open(a.csv) as f:
    for r in csv.DictReader(f):
        nm = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm.split("(")[0].split("<")[0][-48:]))
rows.sort()
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
print(f"{len(rows)} dispatches, span {span / 1e6:.1f} ms, sum of kernel durations {busy / 1e6:.1f} ms")
gaps = collections.defaultdict(lambda: [0, 0.0])
big = []
end = rows[0][1]
prev = rows[0][2]
for s, e, n in rows[1:]:
    g = s - end
    if g > a.min_us * 1e3:
        k = (prev, n)
        gaps[k][0] += 1
        gaps[k][1] += g
        big.append((g, s - rows[0][0], prev, n))
    if e > end:
        end, prev = e, n
tot = sum(v[1] for v in gaps.values())
print(f"idle in gaps > {a.min_us} us: {tot / 1e6:.1f} ms")
for (p, n), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:a.top]:
    print(f"  {t / 1e6:9.2f} ms in {c:5d} gaps (mean {t / c / 1e3:8.1f} us)  after {p}  before {n}")
print("largest gaps:")
for g, at, p, n in sorted(big, reverse=True)[:a.top]:
    print(f"  {g / 1e6:8.2f} ms at +{at / 1e6:9.1f} ms  after {p}  before {n}")

if a.forwards:
    # forwards = [first k_embed_ln* .. next k_pool]; the gap in front of each and its own busy fraction
    fw, cur = [], None
    for s_, e_, n in rows:
        if n.startswith("k_embed_ln"):
            cur = [s_, e_, 0]
        if cur is not None and n.startswith("k_"):
            cur[1] = max(cur[1], e_); cur[2] += e_ - s_
            if n.startswith("k_pool"):
                fw.append(tuple(cur)); cur = None
    print(f"{len(fw)} forwards")
    # runs of back-to-back forwards (gap < 50 ms): per run the number of forwards, wall time, time inside forwards, idle between them
    run = [fw[0]] if fw else []
    def flush(run):
        if len(run) < 8:
            return
        wall = run[-1][1] - run[0][0]
        inside = sum(e_ - s_ for s_, e_, _ in run)
        kern = sum(k for _, _, k in run)
        gaps = [run[i + 1][0] - run[i][1] for i in range(len(run) - 1)]
        print(f"  run of {len(run):4d} forwards at +{(run[0][0] - rows[0][0]) / 1e6:8.1f} ms: wall {wall / 1e6:8.1f} ms, inside forwards {inside / 1e6:8.1f} ms "
              f"(kernel time {kern / 1e6:8.1f}), between forwards {sum(gaps) / 1e6:7.1f} ms (median {sorted(gaps)[len(gaps) // 2] / 1e3:7.1f} us, max {max(gaps) / 1e3:8.1f} us)")
    for f in fw[1:]:
        if f[0] - run[-1][1] > 50e6:
            flush(run); run = []
        run.append(f)
    flush(run)
