cd $GRAFT_REPO_ROOT
for flags in "-DHS_A2_LA=4 -DHS_A2_NOSTORE" "-DHS_A2_LA=4 -DHS_A2_NOSTORE -DHS_A2_NOLDS" "-DHS_A2_LA=6"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -Iinclude -Iholoscene_amd/csrc -c holoscene_amd/csrc/appearance2.hip -o holoscene_amd/csrc/appearance2.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o holoscene_amd/csrc/libholoscene_hip.so holoscene_amd/csrc/*.o
  echo "== $flags"; timeout 200 python tools/exp/appear2_check.py 2>&1 | grep "k_appear2" | tail -1
done
