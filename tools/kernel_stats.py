"""Per-kernel time table from a rocprofv3 rocpd database (or the run's *_kernel_trace.csv):
python tools/kernel_stats.py gpurun_out/prof/b_results.db [steps] -> name, calls, total ms, avg us, % (stdout, CSV)"""
import csv
import glob
import sqlite3
import sys
from collections import defaultdict

path = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = defaultdict(lambda: [0, 0.0])
if path.endswith('.db'):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    cols = [r[1] for r in cur.execute(f'pragma table_info({ks})')]
    namecol = 'display_name' if 'display_name' in cols else 'kernel_name'
    for name, s, e in cur.execute(f'select k.{namecol}, d.start, d.end from {kd} d join {ks} k on d.kernel_id = k.id'):
        a = agg[name]
        a[0] += 1
        a[1] += (e - s) * 1e-6
else:
    for f in glob.glob(path):
        for r in csv.DictReader(open(f)):
            a = agg[r['Kernel_Name']]
            a[0] += 1
            a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
tot = sum(v[1] for v in agg.values())
w = csv.writer(sys.stdout)
w.writerow(['kernel', 'calls', 'total_ms', 'avg_us', 'pct', 'ms_per_step'])
for name, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    w.writerow([name[:110], n, f'{ms:.3f}', f'{ms / n * 1e3:.1f}', f'{100 * ms / tot:.2f}', f'{ms / steps:.3f}'])
print(f'# total kernel time {tot:.1f} ms over {steps:g} steps = {tot / steps:.1f} ms/step (streams overlap: the sum can exceed the step)')
