"""Kernel time of ONE timed bench step (between two consecutive frame_u8 launches of a rocprofv3 --kernel-trace of bench.py),
grouped by kernel family, and the share that is not this repository's code.
usage: python scripts/step_breakdown.py <dir with *_kernel_trace.csv> [out.json]"""
import csv, glob, json, os, re, statistics, sys

f = glob.glob(os.path.join(sys.argv[1], '**', '*_kernel_trace.csv'), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'frame_u8' in r['Kernel_Name']]
OWN = ('ide3d', 'skip_upsample', 'bilinear_up2', 'mapping_kernel')


def family(n):
    n = n.replace('(anonymous namespace)::', '')
    m = re.search(r'modconv_kernel<(\d)', n)
    if m:
        return {'0': 'modconv 3x3', '1': 'modconv 1x1 heads', '2': 'modconv transposed (per class)', '3': 'modconv transposed (all-class)',
                '4': 'modconv 3x3 stride 2'}[m.group(1)]
    m = re.search(r'(\w+_kernel)', n)
    return m.group(1) if m and any(k in n for k in OWN) else ('non-ide3d: ' + re.sub(r'<.*', '', n).replace('void ', '')[:48])


steps = []
for a, b in zip(idx[8:-2], idx[9:-1]):
    fam = {}
    for r in rows[a + 1:b + 1]:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        k = family(r['Kernel_Name'])
        fam[k] = fam.get(k, 0.0) + d
    steps.append(dict(span_us=(int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3, families=fam,
                      launches=b - a, non_ide3d_us=sum(v for k, v in fam.items() if k.startswith('non-ide3d'))))
med = sorted(steps, key=lambda s: s['span_us'])[len(steps) // 2]
out = dict(source=os.path.basename(f), steps_analysed=len(steps), median_step=med,
           median_non_ide3d_us=statistics.median(s['non_ide3d_us'] for s in steps))
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
