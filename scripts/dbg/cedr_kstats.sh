# rocprofv3 kernel stats of the CEDR-KNRM sibling bench (top kernels), builder-side helper: scripts/dbg/cedr_kstats.sh [env assignments]
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pc
env "$@" PYTHONPATH=$GRAFT_REPO_ROOT timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o cedr -- python $GRAFT_REPO_ROOT/scripts/sibling_bench.py --only CEDRKNRM > /dev/null 2>&1
python $GRAFT_REPO_ROOT/scripts/top_kernels.py /tmp/pc 16
