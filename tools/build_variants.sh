# Tuning / profiling builds of the library (run here, before gpurun; build/var travels to the box).
#   fe3prof : streaming front end with per-phase cycle stamps (-DFE3_PROFILE; prints on stderr, blocking)
#   extra -D flags for an ad-hoc variant:  NAME=foo DEFS="-DFE3_X=1" bash tools/build_variants.sh
# Compile-time knobs the sources understand (none of them exists in the default build):
#   front end   -DFE3_PROFILE (phase clocks + the launch's timeline: every workgroup's start / end / CU)  -DFES_STEP_PRIO=0 (no wave priority by step)
#               -DFE3_ABLATE=mask (results invalid: 1 no sparse outputs, 16 no reference-level rows,
#               32 no bb rows)  -DFE3_NW -DFE3_WG_PER_CU -DFE3_CR_EXTRA=n (ring chips)  -DFE3_FORCE_SPW=n (steps per workgroup)
#               -DFE4_PROFILE -DFE4_ABLATE=mask -DFE4_64MSPS (64 Msps through am_k_fe4<32,1,FE4_NW64>)  -DFE2_PROFILING (tile kernel)
#   extraction  -DAM_XPROF (phase clocks + per-workgroup lifetimes, printed when a context is destroyed)
#   chain       -DAM_WALK_DEBUG (the block walk's lane 0 prints hop counts and cycles)  -DAM_MARK_PROF (phase clocks of the marking kernel, printed by four blocks)  -DAM_CB_HEADW -DAM_CB_GROUP
#
#   refinement  -DRS_PROFILE (am_k_refine_seg: us per phase and workgroup, the launch's timeline)  -DRS_WPS=n (waves per SIMD it is compiled for)
# Run every variant on the GPU box under `timeout`: a variant that computes garbage can loop for ever.
set -e
cd "$(dirname "$0")/../gr-air-modes_amd/csrc"
mkdir -p ../../build/var
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fvisibility=hidden -mllvm -amdgpu-sched-strategy=iterative-ilp -shared -I. -I../../include"
SRC="am_kernels.hip am_refine_seg.hip am_fe3.hip am_fe4.hip am_dcblock.hip am_resample.hip am_capi.hip"
if [ -n "$NAME" ]; then
  /opt/rocm/bin/hipcc $FLAGS $DEFS -o ../../build/var/lib_$NAME.so $SRC
else
  /opt/rocm/bin/hipcc $FLAGS -DFE3_PROFILE=1 -o ../../build/var/lib_fe3prof.so $SRC
fi
ls -la ../../build/var
