set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
for sc in "grid10k --spp 20" "glass_metal --spp 200 --depth 100" "final --spp 100" "final --spp 10"; do
DRAY_GRID_BIG_RATIO=0 timeout -k 10 120 python tools/ab_bench.py --variants 3 --rounds 5 --scene $sc --tag "old $sc"
timeout -k 10 120 python tools/ab_bench.py --variants 3 --rounds 5 --scene $sc --tag "new $sc"
done
