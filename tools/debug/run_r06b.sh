cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
PD_LIB_PATH=$GRAFT_REPO_ROOT/partdistillation_amd/libpd_hip_probes.so timeout 900 python -m pytest tools/probes/test_optin_kernels.py -x -q 2>&1 | tail -3
mkdir -p gpurun_out/r06_parity && cp gpurun_out/parity/*.json gpurun_out/r06_parity/ 2>/dev/null
