cd $GRAFT_REPO_ROOT/image-cases-studies_amd/csrc
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-value -Wno-unused-result -Wno-pass-failed"
for U in 2 3 4 6; do
  /opt/rocm/bin/hipcc $BASE -DICS_UPDATE_U=$U -c ics_kernels.hip -o build/ics_kernels.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libics_hip.so build/*.o
  for W in 2 3 4; do
    (cd $GRAFT_REPO_ROOT && ICS_UPDATE_WG_PER_CU=$W python bench.py --no-cpu-baseline --mode nonblind --no-other-mode --steps 200 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('U=$U wg/cu=$W', d['ms_per_step'], d['kernels_ms']['update']['ms'])")
  done
done
