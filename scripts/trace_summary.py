"""Summarise a rocprofv3 kernel_trace.csv by (kernel, grid) — used to pick apart the decode step."""
import csv, collections, sys
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if 'wrs' not in n:
        continue
    short = n.split('(')[0].replace('void ', '').replace('wrs::', '')
    if short.startswith('_ZN3wrs'):
        short = short[7:40]
    key = (short[:34], r['Grid_Size_X'], r['Grid_Size_Y'], r['VGPR_Count'], r['LDS_Block_Size'])
    agg[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = sum(sum(v) for v in agg.values())
print(f"{'kernel':36s} {'grid':>14s} {'vgpr':>5s} {'lds':>6s} {'n':>6s} {'avg_us':>8s} {'med_us':>8s} {'min_us':>8s} {'max_us':>8s} {'tot_ms':>8s} {'%':>5s}")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{k[0]:36s} {k[1]+'x'+k[2]:>14s} {k[3]:>5s} {k[4]:>6s} {len(v):6d} {sum(v)/len(v)/1e3:8.1f} {v[len(v)//2]/1e3:8.1f} {v[0]/1e3:8.1f} {v[-1]/1e3:8.1f} {sum(v)/1e6:8.2f} {100*sum(v)/tot:5.1f}")
