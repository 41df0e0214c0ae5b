#!/bin/bash
# Build a diagnostic variant of libclip_amd.so into gpurun_tmp/ (not the product library):
#   tools/build_variant.sh NAME [-DGEMM8_ABLATE=1 ...]    ->  gpurun_tmp/libNAME.so
# Run with CLIP_AMD_LIB=gpurun_tmp/libNAME.so python tools/gemm_bench.py
set -e
cd "$(dirname "$0")/../clip_amd/csrc"
name=$1; shift
out=../../gpurun_tmp/lib$name.so
mkdir -p ../../gpurun_tmp/obj_$name
for f in gemm gemm_persist gemm8 attention rowwise engine; do
  extra=""; [ $f = attention ] && extra="-mllvm -amdgpu-mfma-vgpr-form"   # as the Makefile
  /opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wno-unused-function $extra "$@" -c $f.hip -o ../../gpurun_tmp/obj_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out ../../gpurun_tmp/obj_$name/*.o
echo built $out
