#!/bin/bash
# the engine with measurement code compiled in, for the lab tools (loaded through PYMC_AMD_LIB; build/ travels to the GPU box):
#   bash tools/build_ticks.sh            -DNUTS_KTIMING  -> build/libnuts_ticks.so     phase timestamps (tools/gb_ticks.py, tree_ticks.py)
#   bash tools/build_ticks.sh knockout   -DNUTS_KNOCKOUT -> build/libnuts_knockout.so  parts of k_rows_gb switched off by NUTS_GA_FLAGS (tools/gb_knockout.py)
export PYMC_AMD_HONOUR_NUTS_ENV=1   # the NUTS_* variables reach the engine as schedule options (nuts_set_option)
KIND=${1:-ticks}
#   bash tools/build_ticks.sh gbw8       -DGB_W=8        -> build/libnuts_gbw8.so      group-block pass with 8 waves per workgroup (A/B; run with NUTS_ROWS_GPW=8)
if [ "$KIND" = knockout ]; then DEF=-DNUTS_KNOCKOUT; elif [ "$KIND" = gbw8 ]; then DEF=-DGB_W=8; else DEF=-DNUTS_KTIMING; fi
cd "$(dirname "$0")/.." && mkdir -p build && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $DEF -Wno-unused-value -Wno-unused-result \
  -Iinclude -Ipymc_amd/csrc -shared -fPIC pymc_amd/csrc/engine.hip -o build/libnuts_$KIND.so
