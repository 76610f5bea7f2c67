#!/usr/bin/env bash
# Build libinstantrestore_hip.so (gfx950 only) in-tree, next to the Python package.
# hipcc cross-compiles without a GPU. Usage: build.sh [extra hipcc flags]
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${IR_OUT:-${HERE}/../libinstantrestore_hip.so}"
BUILD_DIR="${IR_BUILD_DIR:-build}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
SRCS=(linear_tiled.hip shared_attn_fwd.hip shared_attn_fwd_pipe.hip shared_attn_fwd_w64.hip shared_attn_fwd_w128.hip attn_probs.hip adain.hip image_io.hip linear_skinny.hip bench_hooks.hip c_abi.hip)
cd "${HERE}"
# the hand-placed instruction stream of the 128-row attention kernel is generated (committed; regenerated here so it cannot go stale)
python3 "${HERE}/w128/gen.py"
OBJS=()
pids=()
mkdir -p "${BUILD_DIR}"
for s in "${SRCS[@]}"; do
  o="${BUILD_DIR}/${s%.hip}.o"
  OBJS+=("$o")
  extra=()
  # the 64-row kernels need scalar (single-issue) fp32 code where the SLP vectoriser would form v_pk_* operations
  [[ "$s" == shared_attn_fwd_w64.hip ]] && extra+=(-fno-slp-vectorize)
  # the dump kernels exponentiate every MFMA result on the VALU: MFMA destinations in VGPRs (hipcc's default puts them in AGPRs
  # and copies each one out with v_accvgpr_read: 96 of 271 vector instructions per 64 x 64 tile, profiles/r5_pmc_probs.txt)
  [[ "$s" == attn_probs.hip ]] && extra+=(-mllvm -amdgpu-mfma-vgpr-form)
  # round 6: the 128-rows-per-wave kernel (one wave per SIMD, hand-placed stream) is the default wherever the 64-row kernel was and
  # the call is in its domain: same-box sustained A/B at cfg 2's top layer 0.709 vs 0.791 ms (profiles/r6_w128_ab.txt); IR_ATTN_W128=0 restores
  [[ "$s" == shared_attn_fwd.hip ]] && extra+=(-DIR_W128_DEFAULT=${IR_W128_DEFAULT:-1})
  # resource remarks (registers, spills, scratch per kernel) go to <build dir>/<source>.remarks: tools/check_resources.py reads them
  "${HIPCC}" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage "${extra[@]}" "$@" -c "$s" -o "$o" 2> "${BUILD_DIR}/${s%.hip}.remarks" &
  pids+=($!)
done
fail=0
for p in "${pids[@]}"; do wait "$p" || fail=1; done
if [[ $fail != 0 ]]; then grep -h -v "Rpass-analysis" "${BUILD_DIR}"/*.remarks >&2 || true; echo "build.sh: compilation failed" >&2; exit 1; fi
grep -h -E "warning|error" "${BUILD_DIR}"/*.remarks >&2 || true
# kernels that keep asm-issued loads in C++ variables must not spill: fail the build if they do
python3 "${HERE}/../../tools/check_resources.py" "${BUILD_DIR}"
"${HIPCC}" --offload-arch=gfx950 -shared -fPIC "${OBJS[@]}" -o "${OUT}"
echo "built ${OUT}"
