#!/bin/bash
# timing-only ablations / compile-time A-B of the AlignConv and conv-tower kernels on the pyramid launch: every argument is
# the EXTRA flag string of one build of dcn_ops.o ON THE GPU BOX's copy, e.g.
#   bash scripts/abl.sh "-DS2A_ABL=0" "-DS2A_ABL=2"
# S2A_ABL bits: 1 = no epilogue (plain kernel), 2 = loaders skip the blend, 4 = matrix waves skip the MFMAs,
# 8 = loaders skip their corner reads (ring-3 kernel), 16 = matrix waves skip their fragment reads (ring-3 kernel),
# 32 = filter fragments loaded once, 64 = loaded every stage but from a fixed address (plain kernel).
# (Skipping LOADS is not a valid ablation: the compiler deletes the arithmetic that consumes undefined values.)
cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1   # the objects built below carry measurement switches (s2anet_amd/_lib.py refuses them otherwise)
# whatever happens, the box is left with the DEFAULT build (the Makefile does not track EXTRA)
restore() { rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s 2>&1 | grep -E "error" | head -3; }
trap restore EXIT
for a in "$@"; do
  case "$a" in -D*) flags="$a";; *) flags="-DS2A_ABL=$a";; esac
  rm -f s2anet_amd/csrc/dcn_ops.o
  make -C s2anet_amd/csrc -s EXTRA="$flags" 2>&1 | grep -E "error" | head -3
  echo "[$flags] $(timeout -k 10 200 python scripts/bench_pyr.py 2>&1 | grep '"op"' | cut -c1-120 | tr '\n' ' ')"
done
