// Round 5: issue cost of the integer / conversion instructions the Fourier kernel's score phase is made of (dropout hash =
// 2 x v_mul_lo_u32 per element), per wave-instruction in shader cycles, at 1 and 3 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip -o tools/_bin/valu_rate_probe && tools/_bin/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define REP 64
template <int OP>
__global__ __launch_bounds__(256) void probe(uint32_t* out, uint64_t* cyc, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + 1u, b = a ^ 0x9e3779b9u, c = a + 77u, d = b + 99u;
    const uint32_t k1 = 0x85ebca6bu, k2 = 0xc2b2ae35u;
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (OP == 0) { asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(k1)); }
            if (OP == 1) { asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(k1)); }
            if (OP == 2) { asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(k1)); }
            if (OP == 3) { asm volatile("v_mad_u32_u16 %0, %0, %4, %1 op_sel:[1,0,0,0]\n v_mad_u32_u16 %1, %1, %4, %2 op_sel:[0,1,0,0]\n v_mad_u32_u16 %2, %2, %4, %3\n v_mad_u32_u16 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k2)); }
            if (OP == 4) { asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(k1)); }
            if (OP == 5) { asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %0" : "+v"(*(uint64_t*)&a), "+v"(*(uint64_t*)&c)); asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %0" : "+v"(*(uint64_t*)&a), "+v"(*(uint64_t*)&c)); }
            if (OP == 6) { asm volatile("v_cvt_pk_f16_f32 %0, %0, %1\n v_cvt_pk_f16_f32 %1, %1, %2\n v_cvt_pk_f16_f32 %2, %2, %3\n v_cvt_pk_f16_f32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (OP == 7) { asm volatile("v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %1, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %2, %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %3, %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (OP == 8) { asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(k1)); }
            if (OP == 9) { asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %1, %2, %3 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %2, %3, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %3, %0, %1 op_sel_hi:[1,0,0]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (OP == 10) { asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (OP == 11) { asm volatile("v_cvt_f32_f16 %0, %0\n v_cvt_f32_f16 %1, %1\n v_cvt_f32_f16 %2, %2\n v_cvt_f32_f16 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
            if (OP == 12) { asm volatile("v_pk_mul_lo_u16 %0, %0, %4\n v_pk_mul_lo_u16 %1, %1, %4\n v_pk_mul_lo_u16 %2, %2, %4\n v_pk_mul_lo_u16 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k1)); }
            if (OP == 13) { asm volatile("v_lshl_add_u32 %0, %0, 16, %1\n v_lshl_add_u32 %1, %1, 16, %2\n v_lshl_add_u32 %2, %2, 16, %3\n v_lshl_add_u32 %3, %3, 16, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP>
void run(const char* name) {
    uint32_t* out; uint64_t* cyc;
    hipMalloc(&out, 256 * 3072 * 4); hipMalloc(&cyc, 3072 * 8);
    const int iters = 200;
    for (int wps : {1, 2, 3}) {
        const int blocks = 256 * wps;      // 256 threads = 1 wave per SIMD per block
        hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint64_t> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        const double n = (double)iters * REP * 4;
        // s_memtime ticks at 100 MHz; event time -> cycles at 2.4 GHz nominal per SIMD: each SIMD ran wps waves
        printf("%-22s waves/SIMD %d: %7.2f ns per wave-instr per SIMD (= %5.2f cyc @2.4GHz), kernel %.3f ms\n", name, wps,
               ms * 1e6 / (n * wps), ms * 1e6 / (n * wps) * 2.4, ms);
    }
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1>("v_xor_b32"); run<0>("v_mul_lo_u32"); run<8>("v_mul_hi_u32"); run<2>("v_mul_u32_u24"); run<4>("v_mad_u32_u24");
    run<3>("v_mad_u32_u16"); run<12>("v_pk_mul_lo_u16"); run<13>("v_lshl_add_u32"); run<7>("v_xor_b32_sdwa"); run<5>("v_pk_mul_f32");
    run<6>("v_cvt_pk_f16_f32"); run<11>("v_cvt_f32_f16"); run<9>("v_fma_mix_f32"); run<10>("v_fma_f32");
    return 0;
}
