#!/usr/bin/env bash
# Development build of the library WITH the documented experiments (DESIGN/NOTES: kernel variants 1-4, 6, 8, 9, 15, 16, 17 and
# the timing ablations / phase traces): the product sources of instantrestore_amd/csrc plus this directory's
#   shared_attn_fwd_sp.hip   variant 16: one wave per SIMD, spelled-out MFMA/VALU interleave (round 2, negative result)
#   shared_attn_fwd_tp.hip   variant 17: 32 rows per wave, three-stage pipeline (round 2, negative result)
#   shared_attn_fwd_pp.hip   variant  8: ping-pong wave groups, 32 rows per wave (round 1)
#   shared_attn_fwd_w64_dev.hip   the 64-row kernel WITH its PP template branch (variant 15), W64_TRACE / W64_PP_TRACE phase
#                                 stamps and W64_ABL_* timing ablations - replaces csrc/shared_attn_fwd_w64.hip in this build
#   linear_skinny_dev.hip    the X-stationary GEMM with LIN_TRACE / LIN_ABL_* - replaces csrc/linear_skinny.hip
#   linear_xs_pp.hip         round 3: the K = 320 X-stationary GEMM with ping-pong wave groups (ir_linear_fwd_ex kernel id 9;
#                            -DXSPP_TRACE=<block> phase stamps, -DXSPP_ABL=<bits> timing ablations)
#   linear_xs_rot.hip        round 3: the same with a wave's two row blocks rotated, staging / stores between the MFMAs (id 10;
#                            -DXSROT_TRACE=<block>).  Both bit-identical to the product kernel and no faster: the shape runs at
#                            the board power cap (NOTES.md 9.6, profiles/r3_gemm_power.txt)
# None of this is in the product library.  usage: tools/experiments/build.sh [-DW64_TRACE ...]; output: $IR_OUT
# (default tools/experiments/libinstantrestore_hip_dev.so); use it with IR_LIB_PATH=<that file> and the tools/ scripts.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
CSRC="${HERE}/../../instantrestore_amd/csrc"
OUT="${IR_OUT:-${HERE}/libinstantrestore_hip_dev.so}"
BUILD_DIR="${IR_BUILD_DIR:-${HERE}/build}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
PRODUCT=(linear_tiled.hip shared_attn_fwd.hip shared_attn_fwd_pipe.hip attn_probs.hip adain.hip image_io.hip bench_hooks.hip c_abi.hip)
DEV=(shared_attn_fwd_w64_dev.hip linear_skinny_dev.hip linear_xs_pp.hip linear_xs_rot.hip shared_attn_fwd_sp.hip shared_attn_fwd_tp.hip shared_attn_fwd_pp.hip)
mkdir -p "${BUILD_DIR}"
OBJS=(); pids=()
for s in "${PRODUCT[@]}"; do
  o="${BUILD_DIR}/${s%.hip}.o"; OBJS+=("$o")
  "${HIPCC}" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DIR_ABLATIONS -I"${CSRC}" "$@" -c "${CSRC}/$s" -o "$o" &
  pids+=($!)
done
for s in "${DEV[@]}"; do
  o="${BUILD_DIR}/${s%.hip}.o"; OBJS+=("$o")
  extra=()
  [[ "$s" == shared_attn_fwd_w64_dev.hip || "$s" == shared_attn_fwd_sp.hip || "$s" == shared_attn_fwd_tp.hip ]] && extra+=(-fno-slp-vectorize)
  "${HIPCC}" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DIR_ABLATIONS -I"${CSRC}" "${extra[@]}" "$@" -c "${HERE}/$s" -o "$o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
"${HIPCC}" --offload-arch=gfx950 -shared -fPIC "${OBJS[@]}" -o "${OUT}"
echo "built ${OUT}"
