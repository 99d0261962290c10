"""Per-kernel register / spill / occupancy figures of one csrc translation unit, read off the gfx950 ISA hipcc emits
(the .amdhsa_ directives of every kernel): `python tools/isa_stats.py attn_space_bwd.hip [name filter]`.
Also counts v_mfma / ds_read / v_exp instructions per kernel body (static counts)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lavila_amd.build import EXTRA_FLAGS  # noqa: E402


def isa(src):
    out = os.path.join(tempfile.mkdtemp(), 'k.s')
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-gpu-rdc', '-S', '--cuda-device-only',
           *EXTRA_FLAGS.get(os.path.basename(src), []), src, '-o', out]
    subprocess.check_call(cmd)
    return open(out).read()


def main():
    src = sys.argv[1]
    if not os.path.exists(src):
        src = os.path.join(ROOT, 'lavila_amd', 'csrc', src)
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    text = isa(src)
    demangle = lambda n: subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    bodies = {}
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\.amdhsa_kernel \1\n', text, re.S | re.M):
        bodies[m.group(1)] = m.group(2)
    for m in re.finditer(r'\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel', text, re.S):
        name, blk = m.group(1), m.group(2)
        dn = demangle(name)
        if flt and flt not in dn:
            continue
        get = lambda k: (re.search(r'\.amdhsa_' + k + r' (\S+)', blk) or [None, '?'])[1]
        body = bodies.get(name, '')
        cnt = lambda pat: len(re.findall(pat, body, re.M))
        pats = [('v_mfma', r'^\s*v_mfma'), ('ds_read', r'^\s*ds_read'), ('ds_write', r'^\s*ds_write'), ('v_exp', r'^\s*v_exp'),
                ('valu', r'^\s*v_(?!mfma)'), ('salu', r'^\s*s_(?!waitcnt|barrier|nop)'),
                ('global_ld', r'^\s*(global|buffer)_load'), ('global_st', r'^\s*(global|buffer)_store'),
                ('scratch_ops', r'^\s*scratch_')]
        counts = ' '.join(f'{k} {cnt(p)}' for k, p in pats)
        print(f'{dn[:150]}\n    vgpr {get("next_free_vgpr")} (accum_offset {get("accum_offset")}) sgpr {get("next_free_sgpr")} '
              f'scratch {get("private_segment_fixed_size")} lds {get("group_segment_fixed_size")} | {counts}')

if __name__ == '__main__':
    main()
