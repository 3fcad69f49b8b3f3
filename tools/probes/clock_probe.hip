// Which clock does the chip run a bf16 MFMA loop at, as a function of how many CUs run it?  One 8-wave workgroup per CU (a 128 KiB LDS
// request keeps it at one), every wave issues back-to-back v_mfma_f32_32x32x16_bf16 on four independent accumulators; wave 0 of every
// workgroup reads the shader-cycle counter (clock64 = s_memtime) and the 100 MHz wall clock (wall_clock64 = s_memrealtime) around
// the loop.  cycles / wall time = the clock the SIMDs actually ran at; MFMAs x 32 cycles / cycles = how busy the pipe was.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(1024) void mfma_loop(int iters, int fill, unsigned long long* out, float* sink) {
    extern __shared__ char lds[];
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {          // operand bits: zeros (fill 0) or a lane-dependent pattern (fill 1)
        const float va = fill ? (float)((threadIdx.x * 37 + i * 11) % 97) * 0.03f - 1.4f : 0.f;
        const float vb = fill ? (float)((threadIdx.x * 53 + i * 7) % 89) * 0.02f - 0.9f : 0.f;
        a[i] = (__bf16)va; b[i] = (__bf16)vb;
    }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    c1[0] = 1.f; c2[1] = 2.f; c3[2] = (float)threadIdx.x;          // four DIFFERENT chains (identical ones are merged by the compiler)
    __syncthreads();
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = w1 - w0; }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 123.456f) sink[threadIdx.x] = s + lds[threadIdx.x];
}

int main() {
    const int iters = 4000;                 // 16000 MFMAs per wave, two waves per SIMD: ~1 M cycles
    unsigned long long* out; float* sink;
    hipMalloc(&out, 1024 * 16); hipMalloc(&sink, 4096);
    hipFuncSetAttribute((const void*)mfma_loop, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    std::vector<unsigned long long> h(2048);
    printf("operands  waves/SIMD  workgroups   shader cycles    wall us   clock GHz   cycles per MFMA and SIMD   TFLOP/s (all workgroups)\n");
    for (int fill = 1; fill >= 0; --fill)
        for (int wps : {1})          // (in-kernel stamps are wave 0's: with more waves per SIMD the oldest wave keeps its full issue rate and the
            for (int wgs : {16, 48, 96, 192, 256}) {      //  younger ones finish later -- those cases are timed by host events below)
                for (int rep = 0; rep < 3; ++rep) {
                    hipLaunchKernelGGL(mfma_loop, dim3(wgs), dim3(256 * wps), 131072, 0, iters, fill, out, sink);
                    hipDeviceSynchronize();
                }
                hipMemcpy(h.data(), out, wgs * 16, hipMemcpyDeviceToHost);
                std::vector<double> cyc, us;
                for (int i = 0; i < wgs; ++i) { cyc.push_back((double)h[2 * i]); us.push_back((double)h[2 * i + 1] / 100.0); }
                std::sort(cyc.begin(), cyc.end()); std::sort(us.begin(), us.end());
                const double c = cyc[wgs / 2], u = us[wgs / 2];
                const double mfma_per_simd = (double)wps * 4.0 * iters;
                const double flops = (double)wgs * 4.0 * wps * 4.0 * iters * 32768.0;
                printf("%-8s  %10d  %10d   %13.0f   %8.1f   %9.3f   %24.2f   %8.1f\n", fill ? "pattern" : "zeros", wps, wgs, c, u, c / u / 1e3,
                       c / mfma_per_simd, flops / (u * 1e-6) / 1e12);
            }
    // the same through host-side events (no in-kernel counter): 10x the iterations, whole chip
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int fill = 1; fill >= 0; --fill)
        for (int wps : {1, 2, 4}) {
            hipLaunchKernelGGL(mfma_loop, dim3(256), dim3(256 * wps), 131072, 0, iters * 10, fill, out, sink);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(mfma_loop, dim3(256), dim3(256 * wps), 131072, 0, iters * 10, fill, out, sink);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
            const double tf = 256.0 * 4.0 * wps * 4.0 * iters * 10 * 32768.0 / (ms * 1e-3) / 1e12;
            printf("events: %-7s %d waves/SIMD, 256 workgroups, %d MFMAs per wave: %7.3f ms -> %7.1f TFLOP/s = %.3f GHz x 1024 SIMDs x 1024 FLOP/cycle\n",
                   fill ? "pattern" : "zeros", wps, iters * 40, ms, tf, tf * 1e12 / (1024.0 * 1024.0) / 1e9);
        }
    return 0;
}
