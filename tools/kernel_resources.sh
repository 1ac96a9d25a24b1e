#!/bin/bash
# Registers, scratch and LDS of every kernel of the library, from the compiler's own resource remarks (device-only compile of
# each .hip file with the Makefile's flags).  Usage: tools/kernel_resources.sh [file.hip ...] > profiles/rNN_kernel_resources.txt
cd "$(dirname "$0")/../mcmc-date_amd/csrc" || exit 1
files=("$@"); [ ${#files[@]} -eq 0 ] && files=(k_*.hip)
for f in "${files[@]}"; do
  case "$f" in k_mh.hip|k_mh_chain.hip|k_mh_chain_big.hip|k_mh_segment.hip|k_mh_segment_sparse.hip) extra="-mllvm -disable-machine-licm";; *) extra="";; esac   # (the Makefile's NOLICM)
  for g in 0 1 2 3; do
    case "$f" in k_logpdf.hip|k_grad.hip|k_tree_logpdf.hip|k_tree_grad.hip) def="-DMCD_RGROUP=$g";; *) def=""; [ $g -gt 0 ] && continue;; esac
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=off --cuda-device-only $def $extra -c "$f" -o /dev/null \
      -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs:|VGPRs Spill|ScratchSize|LDS Size|Occupancy" \
      | sed 's/.*remark: *//; s/ *\[-Rpass.*//' | paste - - - - - - - | while IFS=$'\t' read -r name v a sc oc sp lds; do
        printf "%-14s %s | %s | %s | %s | %s | %s | %s\n" "$f" "$(echo "${name#Function Name: }" | c++filt | sed 's/(.*//' | cut -c1-70)" "$v" "$a" "$sp" "$sc" "$lds" "$oc"
      done
  done
done
