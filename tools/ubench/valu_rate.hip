// Issue cost of the VALU instructions the epilogues of this library are made of (Philox multiplies, sigmoid), one wave per SIMD and
// four waves per SIMD: cycles per instruction and wave.   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
// MODE 0 v_mul_lo_u32 | 1 v_mul_hi_u32 | 2 v_mad_u64_u32 (lo and hi at once) | 3 v_exp_f32 | 4 v_rcp_f32 | 5 v_xor_b32 | 6 v_fma_f32 | 7 v_mul_u32_u24 | 8 v_pk_fma_f32
template <int MODE>
__global__ void k(uint32_t* out, long long* cyc, int iters) {
    uint32_t x[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 2654435761u + i; f[i] = 1.0f + threadIdx.x * 1e-3f + i; }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(0xD2511F53u));
            if (MODE == 1) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(0xD2511F53u));
            if (MODE == 2) { uint64_t r; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(x[i]), "v"(0xD2511F53u) : "vcc"); x[i] = (uint32_t)r ^ (uint32_t)(r >> 32); }
            if (MODE == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
            if (MODE == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
            if (MODE == 5) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(0xD2511F53u));
            if (MODE == 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(0.999f));
            if (MODE == 7) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(0x511F53u));
        }
    }
    const long long t1 = clock64();
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + __float_as_uint(f[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    uint32_t* out; long long* cyc;
    hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 4096 * 8);
    const int iters = 2000;
    const char* names[8] = {"v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32 (+xor)", "v_exp_f32", "v_rcp_f32", "v_xor_b32", "v_fma_f32", "v_mul_u32_u24"};
    for (int waves = 4; waves <= 16; waves *= 4)
        for (int mode = 0; mode < 8; ++mode) {
            long long h[256];
            for (int rep = 0; rep < 2; ++rep) {
#define L(M) if (mode == M) hipLaunchKernelGGL(k<M>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
                L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7)
                hipDeviceSynchronize();
            }
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double m = 0; for (int i = 0; i < 256; ++i) m += (double)h[i]; m /= 256;
            printf("%-24s waves/SIMD %d: %.2f cycles per instruction per wave\n", names[mode], waves / 4, m / (iters * 8.0));
        }
    return 0;
}
