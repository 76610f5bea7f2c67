#!/usr/bin/env bash
# Development build of the library: the PRODUCT sources of instantrestore_amd/csrc compiled with -DIR_ABLATIONS - nothing
# else.  That switch turns on, inside those same files,
#   shared_attn_fwd.hip        tuning values 1-4, 6, 9 (round 1's straight-line / register-staged kernels) and the `>> 5`
#                              timing-ablation bits of the 32-row kernels (tools/gpu_ablate.py)
#   shared_attn_fwd_w64.hip    tuning values 20-28: the ENERGY / timing ablations of the 64-row QS kernel (template parameter
#                              ABL: no DMA, no barrier, Q fragments read once, no exponentials, no row sums, no conversions;
#                              tools/gpu_energy_probe.py, profiles/r5_energy_budget.txt)
# Round 5 removed the seven forked sources this directory used to hold (*_dev.hip copies of product kernels with trace /
# ablation macros, and the sp / tp / pp / xs_pp / xs_rot experiment kernels whose negative results are recorded in NOTES.md and
# profiles/r1_pp_phase_trace.txt, r2_sp_ablation.txt, r2_kernel_experiments.txt, r3_gemm_power.txt): no second copy of a
# product kernel body is left in the tree, so a product fix cannot miss its twin.
# None of the ablation code is in the product library.  usage: tools/experiments/build.sh [extra hipcc flags]; output: $IR_OUT
# (default tools/experiments/libinstantrestore_hip_dev.so); use it with IR_LIB_PATH=<that file> and the tools/ scripts.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
export IR_OUT="${IR_OUT:-${HERE}/libinstantrestore_hip_dev.so}"
export IR_BUILD_DIR="${IR_BUILD_DIR:-${HERE}/build}"
mkdir -p "${IR_BUILD_DIR}"
exec bash "${HERE}/../../instantrestore_amd/csrc/build.sh" -DIR_ABLATIONS "$@"
