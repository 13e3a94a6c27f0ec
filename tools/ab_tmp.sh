set -e
cd $GRAFT_REPO_ROOT
for l in base items items_f4 items_f8 items_f16; do echo "lib $l"; DRAY_HIP_LIB=tools/ab_libs/$l.so timeout -k 10 200 python tools/fixed_cost.py --spp 250 500 1000; done
echo "lib items_f8 grab 1024";  DRAY_GRAB_SAMPLES=1024 DRAY_HIP_LIB=tools/ab_libs/items_f8.so timeout -k 10 200 python tools/fixed_cost.py --spp 250 500 1000
echo "lib items_f16 grab 1024";  DRAY_GRAB_SAMPLES=1024 DRAY_HIP_LIB=tools/ab_libs/items_f16.so timeout -k 10 200 python tools/fixed_cost.py --spp 250 500 1000
