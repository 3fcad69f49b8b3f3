#!/bin/bash
# libxlxmert_hip_prof.so = the EXPERIMENTAL library (run with XL_EXPERIMENTAL=1 XL_LIB=xlxmert_amd/libxlxmert_hip_prof.so) with gemm_relay.hip compiled -DXL_RELAY_PROFILE (other objects from xlxmert_amd/build)
set -e
cd "$(dirname "$0")/.."
XL_EXPERIMENTAL=1 python -c "from xlxmert_amd.build import build_library; build_library()" > /dev/null
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -Wno-unused-result -DXL_EXPERIMENTAL"
/opt/rocm/bin/hipcc $F -DXL_RELAY_PROFILE -c xlxmert_amd/csrc/gemm_relay.hip -o /tmp/gemm_relay_prof.o
OBJS=$(ls xlxmert_amd/build/exp/*.o | grep -v gemm_relay.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/gemm_relay_prof.o -ldl -o xlxmert_amd/libxlxmert_hip_prof.so
echo built xlxmert_amd/libxlxmert_hip_prof.so
