// Do a matrix wave and a VALU wave of the SAME SIMD overlap?  Workgroup of 8 waves (two per SIMD: wave w and w + 4 share SIMD w & 3).
// Role of the first / second wave of a SIMD: M = back-to-back v_mfma_f32_16x16x4_f32 (8 accumulators), V = v_xor chain (8 chains), T = v_exp_f32, - = idle.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/overlap.hip -o tools/ubench/overlap
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ long long run(int role, int iters, uint32_t& sink) {
    f32x4 acc[8];
    uint32_t x[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0, 0, 0, 0}; x[i] = threadIdx.x * 2654435761u + i; f[i] = 1.0f + threadIdx.x * 1e-3f + i; }
    const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    const long long t0 = clock64();
    if (role == 1) for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    if (role == 2) for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(0xD2511F53u));
    }
    if (role == 3) for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
    }
    if (role == 4) for (int it = 0; it < iters; ++it) {   // one dependent chain
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[0]) : "v"(0xD2511F53u));
    }
    if (role == 5) for (int it = 0; it < iters; ++it) {   // MFMA : VALU = 1 : 4 inside one wave
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(0xD2511F53u));
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[(i + 1) & 7]) : "v"(0xD2511F53u));
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[(i + 2) & 7]) : "v"(0xD2511F53u));
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[(i + 3) & 7]) : "v"(0xD2511F53u));
        }
    }
    if (role == 6) {   // back-to-back v_mfma_f32_16x16x16_bf16
        s16x4 a4 = {(short)0x3f80, (short)0x3f80, (short)0x3f80, (short)0x3f80}, b4 = a4;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
        }
    }
    if (role == 7) {   // back-to-back v_mfma_f32_16x16x32_bf16
        bf16x8 a8, b8;
        for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)1.0f; b8[i] = (__bf16)0.5f; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    f32x4 s = acc[0];
    for (int i = 1; i < 8; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) sink += x[i] + __float_as_uint(f[i]);
    sink += __float_as_uint(s[0] + s[1] + s[2] + s[3]);
    return t1 - t0;
}
// roles[k] = role of the k-th wave of every SIMD (k < wps)
__global__ void k(uint32_t* out, long long* cyc, int iters, int r0, int r1, int r2, int r3) {
    const int wv = threadIdx.x >> 6, kth = wv >> 2;
    const int role = kth == 0 ? r0 : kth == 1 ? r1 : kth == 2 ? r2 : r3;
    uint32_t sink = 0;
    const long long c = run(role, iters, sink);
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + wv] = c;
}
int main() {
    uint32_t* out; long long* cyc;
    hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 256 * 16 * 8);
    const int iters = 2000;
    const char* rn = "-MVTD5BW";   // B = bf16 16x16x16 MFMAs, W = bf16 16x16x32 MFMAs
    const int cases[][4] = {{1,0,0,0},{2,0,0,0},{4,0,0,0},{5,0,0,0},{2,2,0,0},{2,2,2,2},{1,1,0,0},{1,2,0,0},{1,3,0,0},{1,2,2,0},{1,2,2,2},{1,1,2,2},{5,5,0,0},{5,5,5,5},{3,3,3,3},{6,0,0,0},{6,2,0,0},{6,2,2,2},{7,0,0,0},{7,2,0,0},{7,2,2,2},{6,6,0,0}};
    for (auto& c : cases) {
        int wps = 0; for (int i = 0; i < 4; ++i) if (c[i]) wps = i + 1;
        long long h[256 * 16];
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(256), dim3(256 * wps), 0, 0, out, cyc, iters, c[0], c[1], c[2], c[3]); hipDeviceSynchronize(); }
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("roles per SIMD [%c%c%c%c]: cycles per loop instruction and wave:", rn[c[0]], rn[c[1]], rn[c[2]], rn[c[3]]);
        for (int kth = 0; kth < wps; ++kth) {
            double m = 0; for (int b = 0; b < 256; ++b) for (int s = 0; s < 4; ++s) m += (double)h[b * 16 + kth * 4 + s];
            m /= 1024;
            const int per = c[kth] == 5 ? 40 : 8;
            printf("  %c %.2f", rn[c[kth]], m / (iters * (double)per));
        }
        printf("\n");
    }
    return 0;
}
