#!/bin/bash
# Scratch: build libvrterrain.so of git revision $1 into vrenderer_amd/lib/variants/$2/ (for same-box A/B timing)
set -e
REV=$1; NAME=$2
T=$(mktemp -d)
mkdir -p $T/vrenderer_amd/csrc $T/include vrenderer_amd/lib/variants/$NAME
for f in $(git ls-tree --name-only $REV vrenderer_amd/csrc/); do git show $REV:$f > $T/$f; done
git show $REV:include/vrterrain.h > $T/include/vrterrain.h
cd $T/vrenderer_amd/csrc
# per-file flags as that revision's build.py had them
EXTRA=""; if git -C /root/repo show $REV:vrenderer_amd/build.py | grep -q "fno-slp-vectorize"; then EXTRA="-fno-slp-vectorize"; fi
EXTRA_D=""; if git -C /root/repo show $REV:vrenderer_amd/build.py | grep -q '"vr_deferred.hip": \["-fno-slp-vectorize"'; then EXTRA_D="-fno-slp-vectorize"; fi
for f in *.hip; do X=""; [ "$f" = vr_raster.hip ] && X=$EXTRA; [ "$f" = vr_deferred.hip ] && X=$EXTRA_D; /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fvisibility=hidden $X -c $f -o $T/${f%.hip}.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/vrenderer_amd/lib/variants/$NAME/libvrterrain.so $T/*.o -ldl
rm -rf $T
echo built $NAME from $REV
