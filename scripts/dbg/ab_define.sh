# A/B of a compile-time define of ics_kernels.hip (or FILE=...) ON THE GPU BOX: scripts/dbg/ab_define.sh NAME v1 v2 ... -- <command>
# (rebuilds build/ics_kernels.o with -DNAME=v and relinks; the snapshot's library is restored by the next gpurun)
NAME=$1; shift; VALS=(); while [ "$1" != "--" ]; do VALS+=("$1"); shift; done; shift
C=image-cases-studies_amd/csrc
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Wno-unused-value -Wno-unused-result"
for v in "${VALS[@]}"; do
  touch $C/${FILE:-ics_kernels.hip}
  make -C $C CXXFLAGS="$BASE -D$NAME=$v" 2>&1 | tail -2 || { echo "build failed for $v"; continue; }
  echo "== $NAME=$v"; "$@"
done
