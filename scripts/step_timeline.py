"""Timeline of ONE timed bench step from a rocprofv3 --kernel-trace of bench.py: every launch on the critical stream with its start
offset, duration and the idle gap before it (streams shown by queue id).  usage: python scripts/step_timeline.py <trace dir> [out.txt]"""
import csv, glob, os, re, sys

f = glob.glob(os.path.join(sys.argv[1], '**', '*_kernel_trace.csv'), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'frame_u8' in r['Kernel_Name']]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = int(rows[a]['End_Timestamp'])
out = []
busy_until = t0
for r in rows[a + 1:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = re.sub(r'\(anonymous namespace\)::|ide3d::|void ', '', r['Kernel_Name'])
    name = re.sub(r'\(.*', '', name)[:64]
    gap = (s - busy_until) / 1e3
    out.append(f'{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  idle before {gap:6.1f}  q{r.get("Queue_Id", "?")}  grid {r.get("Grid_Size", "?"):>9}  {name}')
    busy_until = max(busy_until, e)
text = '\n'.join(out) + f'\nstep span {(int(rows[b]["End_Timestamp"]) - t0) / 1e3:.1f} us, {b - a} launches'
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(text + '\n')
