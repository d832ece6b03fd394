#!/bin/bash
# developer tool: build libtsg_hip with extra -D flags into tools/_ablate/<name>.so     usage: build_variant.sh name -DX=... [-DY=...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/_ablate/obj_$name
for f in shufflingvideosfortsg_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  extra=""
  tgt=${TSG_VARIANT_SRC:-lstm}
  if [ "$b" = "$tgt" ] || [ ! -f tools/_ablate/obj_$name/$b.o ]; then
    if [ "$b" != "$tgt" ] && [ -f shufflingvideosfortsg_amd/csrc/obj/$b.o ]; then cp shufflingvideosfortsg_amd/csrc/obj/$b.o tools/_ablate/obj_$name/$b.o; continue; fi
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -fno-math-errno "$@" -c $f -o tools/_ablate/obj_$name/$b.o
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC tools/_ablate/obj_$name/*.o -o tools/_ablate/$name.so
echo tools/_ablate/$name.so
