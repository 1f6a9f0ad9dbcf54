"""How do the decode stream and the side stream (next batch's mel / encoder / cross-K/V) share the GPU in overlapped mode?
Reads a rocprofv3 kernel_trace.csv of `bench.py` and reports, per hardware queue, the busy time; the time both queues have a
kernel in flight; and, per kernel of either queue, the average duration while the OTHER queue is busy against while it is not.
    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python3 scripts/overlap_trace.py out/.../t_kernel_trace.csv
"""
import bisect
import collections
import csv
import sys


def short(n):
    s = n.split('(')[0].replace('void ', '').replace('wrs::', '')
    if s.startswith('_ZN3wrs'):
        s = s[7:40]
    return s[:40]


def main(path):
    rows = []
    for r in csv.DictReader(open(path)):
        n = r['Kernel_Name']
        if 'wrs' not in n:
            continue
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], short(n), r['Grid_Size_X']))
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r[2]].append(r)
    qs = sorted(byq, key=lambda q: -len(byq[q]))
    print("queues (by kernel count):")
    for q in qs:
        v = byq[q]
        print(f"  queue {q}: {len(v)} kernels, busy {sum(e - s for s, e, *_ in v) / 1e6:.1f} ms, from {min(s for s, *_ in v) / 1e6:.1f} to {max(e for _, e, *_ in v) / 1e6:.1f} ms")
    if len(qs) < 2:
        return
    dec, side = qs[0], qs[1]
    # only the part of the trace where the side queue is active at all (its first to its last kernel), per contiguous burst
    sv = sorted(byq[side])
    dv = sorted(byq[dec])
    s_starts = [s for s, *_ in sv]
    d_starts = [s for s, *_ in dv]

    def busy_inside(iv_sorted, starts, a, b):
        """time within [a, b) covered by the (non-overlapping, sorted) intervals"""
        i = max(0, bisect.bisect_right(starts, a) - 1)
        t = 0
        while i < len(iv_sorted) and iv_sorted[i][0] < b:
            s, e = iv_sorted[i][0], iv_sorted[i][1]
            t += max(0, min(e, b) - max(s, a))
            i += 1
        return t

    both = sum(busy_inside(sv, s_starts, s, e) for s, e, *_ in dv)
    print(f"decode queue {dec}, side queue {side}: both busy {both / 1e6:.1f} ms")
    side_lo, side_hi = sv[0][0], sv[-1][1]
    for name, mine, other, ostarts in (("decode", dv, sv, s_starts), ("side", sv, dv, d_starts)):
        agg = collections.defaultdict(lambda: [0, 0, 0, 0, 0.0])  # n_alone, t_alone, n_shared, t_shared, shared fraction sum
        for s, e, _, k, g in mine:
            if name == "decode" and not (side_lo <= s <= side_hi):
                continue  # compare like with like: only while the side queue has work queued at all
            ov = busy_inside(other, ostarts, s, e)
            a = agg[(k, g)]
            if ov > 0.2 * (e - s):
                a[2] += 1; a[3] += e - s; a[4] += ov / (e - s)
            else:
                a[0] += 1; a[1] += e - s
        print(f"\n{name} queue kernels: duration while the other queue is busy for > 20 % of the kernel vs not")
        print(f"  {'kernel':42s} {'grid':>10s} {'n_alone':>8s} {'avg_us':>8s} {'n_shared':>8s} {'avg_us':>8s} {'covered':>8s}")
        for (k, g), a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][3]))[:16]:
            al = a[1] / a[0] / 1e3 if a[0] else 0
            sh = a[3] / a[2] / 1e3 if a[2] else 0
            cov = a[4] / a[2] if a[2] else 0
            print(f"  {k:42s} {g:>10s} {a[0]:8d} {al:8.1f} {a[2]:8d} {sh:8.1f} {cov:8.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
