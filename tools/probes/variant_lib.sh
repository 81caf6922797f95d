#!/bin/bash
# Build a variant of the library with extra compiler flags for ONE source file (same-box A/B runs: copy it over pathpyg_amd/lib/libpathpyg_amd.so
# on the GPU box).   usage: bash tools/probes/variant_lib.sh NAME "-DFLAG=.." [pp_gcn_wide]   ->  tools/probes/_bin/lib_NAME.so
set -e
cd "$(dirname "$0")/../.."
F=${3:-pp_gcn_wide}
python -c "import __graft_entry__ as g; g.build()" > /dev/null
OBJ=pathpyg_amd/csrc/build
mkdir -p tools/probes/_bin
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $2 -c pathpyg_amd/csrc/$F.hip -o /tmp/${F}_$1.o
objs=$(ls $OBJ/*.o | grep -v "/$F.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/_bin/lib_$1.so $objs /tmp/${F}_$1.o
ls -la tools/probes/_bin/lib_$1.so
