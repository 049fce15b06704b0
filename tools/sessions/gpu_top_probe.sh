#!/bin/bash
# phase clock of k_panel_top (tools/top_probe.cpp): build with hipcc (cross-compiles here), run on the GPU box
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRC5_TIME top_probe.cpp -o top_probe && ./top_probe ${1:-200}
