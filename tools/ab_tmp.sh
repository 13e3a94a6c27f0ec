set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/ab_bench.py --variants 3 --rounds 3 --scene grid10k --width 3840 --spp 256 --tag config4
timeout -k 10 200 python tools/ab_bench.py --variants 3 --rounds 3 --scene glass_metal --spp 1000 --depth 100 --tag config5
timeout -k 10 200 python tools/ab_bench.py --variants 1 --rounds 3 --scene three --width 1920 --spp 100 --tag config2
timeout -k 10 200 python tools/ab_bench.py --variants 1 --rounds 3 --scene three --width 400 --spp 10 --depth 10 --tag config1
