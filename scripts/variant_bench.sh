# Rebuilds one translation unit of libics_hip.so with extra -D flags on the GPU box and runs the blind bench:
#   bash scripts/variant_bench.sh ics_conv_mfma "-DICS_EPI_LOAD_AUX=0" ["more flags" ...]
cd $GRAFT_REPO_ROOT/image-cases-studies_amd/csrc
TU=$1; shift
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-value -Wno-unused-result -Wno-pass-failed"
for FL in "$@"; do
  /opt/rocm/bin/hipcc $BASE $FL -c $TU.hip -o build/$TU.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libics_hip.so build/*.o || { echo "build failed: $FL"; continue; }
  for rep in 1 2; do
    (cd $GRAFT_REPO_ROOT && python bench.py --no-cpu-baseline --steps 200 --warmup 20 --no-other-mode | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$FL', d['ms_per_step'], {k:v['ms'] for k,v in d['kernels_ms'].items()})")
  done
done
