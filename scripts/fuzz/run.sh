#!/bin/bash
# CPU-only robustness run of the untrusted-input paths of libhalart.so: the glTF / PNG / JPEG / PNM / TGA / OpenEXR / .hdr / .pfm loaders compiled with g++
# -fsanitize=address,undefined (GPU sanitizers are not available on this pool) and fed ~6300 mutated files.  Any sanitizer report aborts.
# usage: bash scripts/fuzz/run.sh [workdir]        (result line per corpus; profiles/r02_fuzz.txt holds the last run)
set -eu
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=${1:-/tmp/halart_fuzz}
mkdir -p $W
C=$ROOT/hala-renderer_amd/csrc
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$C \
    $ROOT/scripts/fuzz/harness.cpp $C/gltf_loader.cpp $C/jpeg_decode.cpp $C/host_util.cpp -L/opt/rocm/lib -lamdhip64 -lz -ldl \
    -Wl,-rpath,/opt/rocm/lib -o $W/harness
python3 $ROOT/scripts/fuzz/make_corpus.py $W/corpus 2>/dev/null
$W/harness jpeg $W/corpus/jpeg/*.jpg
$W/harness gltf $W/corpus/gltf/*.gltf
$W/harness gltf $W/corpus/png/*.gltf
$W/harness image $W/corpus/image/*
echo "no sanitizer report"
