# Tuning builds of the library for tools/gpu_variants.sh (run here, before gpurun; build/var travels to the box).
#   nocse       : default geometry, P5 recomputes the in-chip prefix sums (168 -> 143 VGPRs at 64 Msps)
#   nt384       : two 384-thread workgroups per CU (needs <= 128 VGPRs: FE2_WPS=4 + FE2_NO_PREFIX_CSE=1,
#                 6 spilled dwords in the interior body instead of 72)
#   nt384_wps3  : 384 threads, 168-VGPR budget (does not co-reside: control)
set -e
cd "$(dirname "$0")/../gr-air-modes_amd/csrc"
mkdir -p ../../build/var
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -shared -I. -I../../include"
SRC="am_kernels.hip am_fe2.hip am_dcblock.hip am_capi.hip"
/opt/rocm/bin/hipcc $FLAGS -DFE2_NO_PREFIX_CSE=1 -o ../../build/var/lib_nocse.so $SRC
/opt/rocm/bin/hipcc $FLAGS -DFE2_NT=384 -DFE2_WPS=4 -DFE2_NO_PREFIX_CSE=1 -o ../../build/var/lib_nt384.so $SRC
/opt/rocm/bin/hipcc $FLAGS -DFE2_NT=384 -DFE2_WPS=3 -o ../../build/var/lib_nt384_wps3.so $SRC
ls -la ../../build/var
