#!/bin/bash
# usage: variant.sh NAME "-DFLAGS"  -> tools/probes/_bin/lib_NAME.so (only pp_gcn_wide.hip rebuilt with the flags)
set -e
cd /root/repo
python -c "import __graft_entry__ as g; g.build()" > /dev/null
OBJ=pathpyg_amd/csrc/build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $2 -c pathpyg_amd/csrc/pp_gcn_wide.hip -o /tmp/wide_$1.o
objs=$(ls $OBJ/*.o | grep -v pp_gcn_wide.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/_bin/lib_$1.so $objs /tmp/wide_$1.o
ls -la tools/probes/_bin/lib_$1.so
