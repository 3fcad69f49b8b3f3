cd /root/repo; mkdir -p gpurun_out
for s in "16384 3072 768 1 1 6" "1280 3072 768 1 1 6" "16384 3072 768 1 0 7" "1280 3072 768 1 0 7" "16384 2304 768 1 1 0" "16384 768 768 1 1 2" "16384 768 3072 1 1 2" "16384 768 768 1 0 0" "16384 3072 768 1 1 0"; do echo "== $s"; python tools/gemm_trace.py $s 2>&1 | grep -v amdgpu.ids; done > gpurun_out/trace_epi.txt
