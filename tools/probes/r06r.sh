# node attention kernels alone (tools/kernel_bench.py --only node): the forms against each other
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06r; mkdir -p $O
cd $R
run() { echo "== $*"; env "$@" python tools/kernel_bench.py --only node --B $B --N $N --iters 50 2>/dev/null | tail -1; }
for cfg in "256 32" "128 48"; do
  set -- $cfg; B=$1; N=$2
  echo "#### B=$B N=$N"
  run TGT_NODE_KB=0 TGT_NODE_MFMA16=1
  run TGT_NODE_KB=0 TGT_NODE_MFMA16=2
  run TGT_NODE_KB=2 TGT_NODE_KB_HW=64
  run TGT_NODE_KB=2 TGT_NODE_KB_HW=32
done 2>&1 | tee $O/node_kernels.txt
