#!/bin/bash
# Build a diagnostic variant of libclip_amd.so into gpurun_tmp/ (not the product library):
#   tools/build_variant.sh NAME [-DGEMM8_ABLATE=1 ...]    ->  gpurun_tmp/libNAME.so
# Run with CLIP_AMD_LIB=gpurun_tmp/libNAME.so python tools/gemm_bench.py
set -e
cd "$(dirname "$0")/../clip_amd/csrc"
name=$1; shift
out=../../gpurun_tmp/lib$name.so
mkdir -p ../../gpurun_tmp/obj_$name
for f in gemm gemm8 gemm4 attention rowwise; do       # both operand sets, as the Makefile
  extra=""; [ $f = attention ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
  /opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wno-unused-function $extra "$@" -c $f.hip -o ../../gpurun_tmp/obj_$name/$f.o &
  /opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wno-unused-function $extra "$@" -DCLIP_OPERAND_F16=1 -c $f.hip -o ../../gpurun_tmp/obj_$name/${f}_f16.o &
done
for f in engine peak; do
  /opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c $f.hip -o ../../gpurun_tmp/obj_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out ../../gpurun_tmp/obj_$name/*.o
echo built $out
