#!/bin/bash
# Build libsegnb_hip.so for gfx950 (cross-compiles without a GPU).  Output lands next to the sources so
# it travels with the repo snapshot to the GPU box.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -fno-gpu-rdc"
mkdir -p obj
pids=()
for f in conv_igemm fprop_s1 fprop_dma fprop_rw fprop_roll fprop_c8 fprop_sx fprop_thin wgrad_s1 wgrad_roll norm_act head_loss tiles runtime; do
  if [ ! -f obj/$f.o ] || [ $f.hip -nt obj/$f.o ] || [ common.h -nt obj/$f.o ] || [ fprop_dma.h -nt obj/$f.o ] || [ ../../include/segnb_hip.h -nt obj/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o obj/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC obj/conv_igemm.o obj/fprop_s1.o obj/fprop_dma.o obj/fprop_rw.o obj/fprop_roll.o obj/fprop_c8.o obj/fprop_sx.o obj/fprop_thin.o obj/wgrad_s1.o obj/wgrad_roll.o obj/norm_act.o obj/head_loss.o obj/tiles.o obj/runtime.o -o libsegnb_hip.so
echo "built $(pwd)/libsegnb_hip.so"
