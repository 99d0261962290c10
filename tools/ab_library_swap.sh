#!/bin/bash
# Same-box A/B of two builds of liblavila_hip.so (boxes of the pool differ by +-1 %, two runs on one box by +-0.05 %).
#
#   here (build container):  tools/ab_library_swap.sh build <git-commit>     # builds that commit's csrc into
#                                                                             # tools/probes/ab/liblavila_hip_base.so
#   on the GPU box (inside a gpurun command):
#                            tools/ab_library_swap.sh run <out-file> [bench args...]
#       alternates base / new / base / new: copies the library over lavila_amd/lib/liblavila_hip.so (the C ABI must be the
#       same on both sides), runs `python bench.py --no-cpu-baseline --no-events <bench args>` and appends
#       "<which> <pairs/s> <ms per step>" to <out-file>; restores the new library at the end. AB_BASE_ENV="VAR=value ..."
#       is added to the environment of the base runs (a switch the host side needs for an older C ABI).
# The base library is git-ignored (*.so) but travels with the gpurun snapshot; delete tools/probes/ab afterwards.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
L=$ROOT/lavila_amd/lib/liblavila_hip.so
case "${1:-}" in
  build)
    commit=${2:?commit}
    tmp=$(mktemp -d)
    git -C "$ROOT" archive "$commit" lavila_amd/csrc lavila_amd/build.py include | tar -x -C "$tmp"
    touch "$tmp/lavila_amd/__init__.py"
    mkdir -p "$ROOT/tools/probes/ab"
    for f in "$tmp"/lavila_amd/csrc/*.hip; do
      # that commit's own per-file flags (lavila_amd/build.py EXTRA_FLAGS)
      extra=$(cd "$tmp" && python -c "import sys; sys.path.insert(0, '.'); from lavila_amd.build import EXTRA_FLAGS as E; print(' '.join(E.get('$(basename "$f")', [])))")
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -fvisibility=hidden $extra -c "$f" \
        -o "$tmp/$(basename "$f" .hip).o" 2>/dev/null &
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/probes/ab/liblavila_hip_base.so" "$tmp"/*.o
    rm -rf "$tmp"
    ls -la "$ROOT/tools/probes/ab/liblavila_hip_base.so"
    ;;
  run)
    out=${2:?out-file}; shift 2
    cp "$L" /tmp/lavila_new.so
    for v in base new base new; do
      if [ $v = base ]; then cp "$ROOT/tools/probes/ab/liblavila_hip_base.so" "$L"; else cp /tmp/lavila_new.so "$L"; fi
      extra_env=""; [ $v = base ] && extra_env="${AB_BASE_ENV:-}"
      echo "$v $(cd "$ROOT" && env $extra_env timeout 300 python bench.py --no-cpu-baseline --no-events "$@" 2>/dev/null | grep '^{' | \
        python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')" >> "$out"
    done
    cp /tmp/lavila_new.so "$L"
    ;;
  *) sed -n 2,14p "$0"; exit 2;;
esac
