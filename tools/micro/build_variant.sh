# A diagnostic build of the library with extra compiler flags on ONE source file:
#   tools/micro/build_variant.sh NAME file.hip "-DFLAG ..."   ->  tools/micro/lib_NAME.so   (the other objects come from the regular build)
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; src=$2; shift 2
obj=/tmp/variant_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I "$root/diffusion-model_amd/csrc" -I "$root/include" "$@" -c "$root/diffusion-model_amd/csrc/$src" -o $obj
others=$(ls "$root"/diffusion-model_amd/build/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/micro/lib_$name.so" $obj $others
echo "$root/tools/micro/lib_$name.so"
