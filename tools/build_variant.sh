#!/bin/bash
# Developer aid: build an alternative libria_gpu.so with extra -D flags into build/ab/<name>.so (A/B runs via RIA_GPU_LIB).
name=$1; shift
mkdir -p "$(dirname "$0")/../build/ab"
cd "$(dirname "$0")/../ria_amd/csrc" && hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -fPIC -shared -std=c++17 -Wno-inline-asm -Wno-pass-failed "$@" -o ../../build/ab/$name.so ria_gpu.hip && echo built $name
