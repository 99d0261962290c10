"""Builds lavila_amd/lib/liblavila_hip.so from lavila_amd/csrc/*.hip with hipcc for gfx950.

The library depends only on the HIP runtime (no torch, no python): it is the C-ABI drop-in boundary
declared in include/lavila_hip.h. hipcc cross-compiles without a GPU, so this runs in the build
container; the resulting .so travels to the GPU box with the source tree.
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'liblavila_hip.so')
ARCH = 'gfx950'


# Per-file flags.
# -fno-honor-nans for the MFMA attention kernels: without it every fmaxf on an MFMA result gets a canonicalising v_max x,x in
# front (a quarter of the softmax's VALU instructions); their scores are never NaN (finite operands, -inf only through the
# masks, every inf - inf guarded by a select) -- attn_mfma_common.h, max3_raw.
# -fno-slp-vectorize where it measured faster (round 6, profiles/r06_valu_rates.txt): the SLP vectoriser pairs f32 multiplies /
# adds / fmas into v_pk_{mul,add,fma}_f32. Their THROUGHPUT is fine (2.7-3.4 cycles per pair against 1.6-2.2 per scalar
# instruction) but in the dependent chains of a two-waves-per-SIMD kernel they are slower than the scalar forms: the TN GEMM's
# QuickGELU epilogues (+30 % in tools/probes/valu_gelu.hip) and the fused space-attention backward (-3.2 % without them);
# LayerNorm forward (+20 % WITHOUT them), the time attention and the streaming kernels keep the default.
_NO_SLP = ['-fno-slp-vectorize']
EXTRA_FLAGS = {'attn_space_mfma.hip': ['-fno-honor-nans'], 'attn_space_stream.hip': ['-fno-honor-nans'],
               'gemm_tn_mfma.hip': _NO_SLP, 'attn_space_bwd.hip': _NO_SLP}


def _hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (needed to build liblavila_hip.so)')


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')) + \
        [os.path.join(HERE, '..', 'include', 'lavila_hip.h')]


def _digest(srcs=None):
    """Content hash of the sources (file NAMES, not paths: the stamp of one checkout is valid in another)."""
    h = hashlib.sha256()
    for p in (sources() if srcs is None else srcs) + _headers():
        with open(p, 'rb') as f:
            h.update(os.path.basename(p).encode() + b'\0' + f.read())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, 'build.stamp')
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    stamps = {}
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        # per-object stamp: a translation unit is recompiled when it, a header or the flags changed
        odig, ostamp = _digest([src]), obj + '.stamp'
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read().strip() == odig:
            continue
        stamps[ostamp] = odig
        cmd = [hipcc, f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-fvisibility=hidden',
               '-Wall', '-Wno-unused-function'] + EXTRA_FLAGS.get(os.path.basename(src), []) + ['-c', src, '-o', obj]
        if verbose:
            print('[lavila_amd.build]', ' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out.strip() and (verbose or p.returncode):
            print(out)
        if p.returncode:
            failed = True
            print(f'[lavila_amd.build] FAILED: {src}', file=sys.stderr)
    if failed:
        raise RuntimeError('hipcc failed')
    for ostamp, odig in stamps.items():
        with open(ostamp, 'w') as f:
            f.write(odig)
    cmd = [hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print('[lavila_amd.build]', ' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, 'w') as f:
        f.write(dig)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
