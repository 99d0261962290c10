"""Driver: which ONE poison tensor of tools/probe_graph_step_poison.py breaks the replay? Bisects PROBE_FILL_SET over fresh
processes (same allocation sequence every time: all poison tensors are allocated, only a subset is filled), then describes
the memory around the culprit."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env0 = dict(os.environ, PROBE_POISON_ITS='2', PROBE_POISON_STREAMS='cur')


def broken(lo, hi, extra=None):
    env = dict(env0, PROBE_FILL_SET=f'{lo}:{hi}', **(extra or {}))
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'probe_graph_step_poison.py')], env=env,
                         capture_output=True, text=True, timeout=300).stdout
    last = [ln for ln in out.splitlines() if ln.startswith('max |dp|')]
    bad = bool(last) and ('nan' in last[-1].split('nan in poisoned params')[0] or float(last[-1].split()[2]) > 1e-6)
    print(f'fill [{lo},{hi}): {last[-1] if last else "no result"} -> {"BROKEN" if bad else "clean"}', flush=True)
    return bad, out


lo, hi = 0, 96
ok, _ = broken(lo, hi)
if not ok:
    print('the full set does not break the replay in this process layout: nothing to bisect')
    sys.exit(0)
while hi - lo > 1:
    mid = (lo + hi) // 2
    if broken(lo, mid)[0]:
        hi = mid
    elif broken(mid, hi)[0]:
        lo = mid
    else:
        print(f'neither half of [{lo},{hi}) alone breaks it: more than one tensor involved')
        break
print(f'culprit range [{lo},{hi})')
_, out = broken(lo, hi, {'PROBE_DESCRIBE': '1'})
print('\n'.join(ln[:400] for ln in out.splitlines() if ln.startswith('[describe]')))
