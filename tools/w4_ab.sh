set -x
cd $GRAFT_REPO_ROOT
export CLIPX_LIB=libclipx_ablate.so
TAG=${1:-r06x}
shift
CFGS=${@:-3 6 3:16 6:16}
timeout 400 tools/gemm_bench -r 6 -b 30 65536,3072,1024,23 65536,4096,1024,17 65536,1024,1024,6 65536,1024,4096,6 65792,3072,1024,23 -- $CFGS > gpurun_out/${TAG}_w4_ab.log 2>&1
tail -70 gpurun_out/${TAG}_w4_ab.log
