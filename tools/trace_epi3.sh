# K-loop section cycles (profile build) against the number of busy CUs: clock or contention?
cd /root/repo; mkdir -p gpurun_out
export XL_GEMM_DUO=0 XL_LIB=$PWD/xlxmert_amd/libxlxmert_hip_prof.so
for s in "1024 3072 768 1 1 0" "2048 3072 768 1 1 0" "4096 3072 768 1 1 0" "5376 3072 768 1 1 0" "16384 3072 768 1 1 0"; do echo "== $s"; python tools/gemm_trace.py $s 2>&1 | grep -v amdgpu.ids; done > gpurun_out/trace_epi3.txt
