#!/bin/bash
# Developer tool: build libmopt_hip with extra -D flags into ab/<name>.so for same-box A/B runs (MOPT_HIP_LIB=ab/<name>.so).
set -e
name=$1; shift
mkdir -p ab
hipcc -O3 -std=c++17 -shared -fPIC --offload-arch=gfx950 -Wno-unused-value -Iinclude "$@" multioptpy_amd/csrc/*.hip -o ab/$name.so
echo built ab/$name.so
