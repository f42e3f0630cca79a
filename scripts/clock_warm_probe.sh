cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1
trap 'rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s' EXIT
rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s EXTRA=-DS2A_STAMP=1 2>&1 | grep error
for w in 0.0 0.05 0.5 3.0; do S2A_WARM_S=$w timeout -k 10 200 python scripts/stamps_pyr.py 2>&1 | grep "data:\|clock" | cut -c1-150; done
S2A_WARM_S=3.0 timeout -k 10 200 python scripts/stamps_pyr.py zeros 2>&1 | grep "data:\|clock" | cut -c1-150
