set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${TAG:-r06n}; mkdir -p $O
export TMPDIR=/tmp
cd $R
tools/pmc_passes.sh $O "tricol node agg proj" > $O/pmc_passes.log 2>&1; tail -3 $O/pmc_passes.log
PMC_KB_ARGS="--B 128 --N 48" tools/pmc_passes.sh $O/n48 "tricol node" > $O/pmc_passes_n48.log 2>&1; tail -3 $O/pmc_passes_n48.log
bash tools/pmc_edge.sh $O > $O/pmc_edge_summary.txt 2>&1; cat $O/pmc_edge_summary.txt
python -c "
import json; d=json.load(open('$O/pmc_summary.json')); print({k:(v.get('hbm_bytes_per_launch'), v.get('mfma_util')) for k,v in d.items() if isinstance(v,dict)}, d.get('_kernel_src_sha'))"
