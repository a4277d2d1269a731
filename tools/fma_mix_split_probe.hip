#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void k(const float* a, const float* m, uint32_t* out) {
    float a0 = a[threadIdx.x], a1 = a[threadIdx.x + 64], m0 = m[threadIdx.x], m1 = m[threadIdx.x + 64];
    uint32_t h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a0), "v"(m0));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(a1), "v"(m1));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a0), "v"(m0), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(a1), "v"(m1), "v"(h));
    out[threadIdx.x] = h; out[threadIdx.x + 64] = l;
}
int main() {
    float ha[128], hm[128]; uint32_t ho[128];
    // normal values, zeros, and values whose fp16 residual (and, for the last quarter, whose fp16 head) is SUBNORMAL in fp16:
    // v_cvt_pk_f16_f32 produces gradual underflow; the mix instructions must too for a drop-in replacement
    for (int i = 0; i < 128; ++i) {
        ha[i] = 1.2345678f * (i + 1) * (i % 3 == 0 ? -1 : 1); hm[i] = (i % 5 == 0) ? 0.f : 4.f;
        if (i % 4 == 1) { ha[i] = (1.f + (i + 1) * 0x1p-21f) * (i % 8 == 1 ? 1.f : 0x1p-6f); hm[i] = 1.f; }      // residual ~2^-20 .. 2^-27
        if (i % 4 == 2) { ha[i] = (3.f + i) * 0x1p-20f; hm[i] = 0x1p-2f; }                                          // head itself subnormal
    }
    float *a, *m; uint32_t* o;
    hipMalloc(&a, 512); hipMalloc(&m, 512); hipMalloc(&o, 512);
    hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(m, hm, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, m, o);
    hipMemcpy(ho, o, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        for (int half = 0; half < 2; ++half) {
            float x = (half ? ha[i + 64] * hm[i + 64] : ha[i] * hm[i]);
            _Float16 h0 = (_Float16)x; _Float16 h1 = (_Float16)(x - (float)h0);
            uint16_t gh = (ho[i] >> (16 * half)) & 0xffff, gl = (ho[i + 64] >> (16 * half)) & 0xffff;
            uint16_t eh, el; __builtin_memcpy(&eh, &h0, 2); __builtin_memcpy(&el, &h1, 2);
            if (gh != eh || gl != el) { if (bad < 5) printf("lane %d half %d: got %04x %04x want %04x %04x (x=%g)\n", i, half, gh, gl, eh, el, x); ++bad; }
        }
    }
    printf("fma_mix split: %d mismatches of 128\n", bad);
    return bad != 0;
}
