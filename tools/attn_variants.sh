#!/bin/bash
# usage (GPU box): bash tools/attn_variants.sh "<flags of variant 1>" "<flags of variant 2>" ...
# Rebuilds csrc/attn.hip with each set of -DVC_ATTN_* switches, checks the dense-attention tests and times the kernel
# (tools/attn_one.py at B = 64, S = 577 / 578).  The library is restored to the default build at the end.
R=$GRAFT_REPO_ROOT
cd $R
for V in "$@" ""; do
  touch vitcap_amd/csrc/attn.hip
  make -C vitcap_amd/csrc EXTRA="$V" 2>&1 | grep -E "error|spill" | head -5
  echo "=== variant [$V]"
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes vitcap_amd/libvitcap_hip.so 2>/dev/null | grep -A12 "attn_dense_kernelILb0" | grep -E "vgpr_count|vgpr_spill|group_segment_fixed" | tr '\n' ' '; echo
  if [ -z "$NOTEST" ]; then python -m pytest tests/test_hip_ops.py -k "attn_dense" -x -q 2>&1 | tail -1; fi
  python tools/attn_one.py 64 577 30 2>&1 | grep attn_dense
  python tools/attn_one.py 64 578 30 2>&1 | grep attn_dense
done
