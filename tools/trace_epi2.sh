# epilogue time against the number of concurrently running whole-CU tiles (is the epilogue bound by a shared resource?)
cd /root/repo; mkdir -p gpurun_out
export XL_GEMM_DUO=0
for s in "1024 3072 768 1 1 6" "2048 3072 768 1 1 6" "4096 3072 768 1 1 6" "5376 3072 768 1 1 6" "1024 3072 768 1 1 0" "4096 3072 768 1 1 0" "5376 3072 768 1 1 0" "1024 3072 768 1 0 7" "5376 3072 768 1 0 7"; do echo "== $s"; python tools/gemm_trace.py $s 2>&1 | grep -v amdgpu.ids; done > gpurun_out/trace_epi2.txt
