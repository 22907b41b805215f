#!/bin/bash
# Development aid: build libmtsv_amd_<name>.so with extra -D flags on ONE device file, next to the product library,
# so that one GPU call can time several variants of a kernel back to back:
#     tools/build_variant.sh dec16 k_verify "-DMTSV_SW_DECIDE=16"
#     MTSV_AMD_LIB=mtsv_tools_amd/libmtsv_amd_dec16.so python bench.py --resident-only
# (the variant libraries are git-ignored; the spill guard runs on them like on the product build)
set -e
name=$1; file=$2; flags=$3
cd "$(dirname "$0")/../mtsv_tools_amd/csrc"
make -s -j8 >/dev/null
mkdir -p build/var
# (file may be a comma-separated list of .hip files)
for f in ${file//,/ }; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result --offload-arch=gfx950 -ffp-contract=off \
        -fno-slp-vectorize $flags -Rpass-analysis=kernel-resource-usage -c $f.hip -o build/var/${f}_$name.o 2> build/var/${f}_$name.remarks
    python3 ../../tools/kernel_resources.py build/var/${f}_$name.remarks --guard > build/var/${f}_$name.resources.txt || { grep -v "spill   0 vgpr_spill   0" build/var/${f}_$name.resources.txt; exit 1; }
done
objs=""
for o in mgindex builder capi dev_index k_seed k_coalesce k_verify batch gpu_builder host_pack; do
    case ",$file," in *",$o,"*) objs="$objs build/var/${o}_$name.o";; *) objs="$objs build/$o.o";; esac
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libmtsv_amd_$name.so $objs -lpthread
echo built mtsv_tools_amd/libmtsv_amd_$name.so
