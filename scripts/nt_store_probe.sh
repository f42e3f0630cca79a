cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1
trap 'rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s' EXIT
for fl in "-DS2A_DCN_NT_STORE=0" "-DS2A_DCN_NT_STORE=1"; do
  rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s EXTRA="-DS2A_STAMP=1 $fl" 2>&1 | grep error
  echo "== $fl"; timeout -k 10 200 python scripts/stamps_pyr.py zeros 2>&1 | grep "matrix wave 0\|us per launch" | cut -c1-200
done
bash scripts/ab_dcn_build.sh "-DS2A_DCN_NT_STORE=0" "-DS2A_DCN_NT_STORE=1" 2>&1 | grep -v "relu-sparse\|alignconv_fused"
