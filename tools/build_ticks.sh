#!/bin/bash
# the engine with phase timestamps compiled in (NUTS_KTIMING): scratch/libnuts_ticks.so, for tools/gb_ticks.py / tree_ticks.py
export PYMC_AMD_HONOUR_NUTS_ENV=1   # the NUTS_* variables below reach the engine as schedule options (nuts_set_option)
cd "$(dirname "$0")/.." && mkdir -p scratch && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNUTS_KTIMING -Wno-unused-value -Wno-unused-result \
  -Iinclude -Ipymc_amd/csrc -shared -fPIC pymc_amd/csrc/engine.hip -o scratch/libnuts_ticks.so
