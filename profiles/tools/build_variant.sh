#!/bin/bash
# build_variant.sh <name> [source-dir] [extra hipcc flags...]: one more build of libdbgk.so under dbg_assembly_amd/_variants/ (git-ignored,
# travels to the GPU box) for A/B runs inside one gpurun call (profiles/ab_bench.sh, DBGK_LIB)
set -e
R=/root/repo
name=$1; src=${2:-$R/dbg_assembly_amd/csrc}; [ -z "$src" ] && src=$R/dbg_assembly_amd/csrc; shift; shift || true
O=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$src -Wall -Wno-unused-result "$@" -c $src/dbgk.hip -o $O/dbgk.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$src -c $src/dbgk_sort.hip -o $O/dbgk_sort.o
g++ -O3 -std=c++17 -fPIC -I$R/include -Wall -c $src/dbgk_pack.cpp -o $O/dbgk_pack.o
mkdir -p $R/dbg_assembly_amd/_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/dbg_assembly_amd/_variants/$name.so $O/dbgk.o $O/dbgk_sort.o $O/dbgk_pack.o -lpthread
rm -rf $O
echo built $R/dbg_assembly_amd/_variants/$name.so
