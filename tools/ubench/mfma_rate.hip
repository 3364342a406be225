// Issue rate of the MFMA shapes this library uses, on one SIMD (one wave) and on a full CU (4 waves): cycles per instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    s16x4 a4 = {(short)0x3f80, (short)0x3f80, (short)0x3f80, (short)0x3f80}, b4 = a4;
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)1.0f; b8[i] = (__bf16)0.5f; }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
            if (MODE == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    f32x4 s = acc[0];
    for (int i = 1; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096 * 8);
    const int iters = 2000;
    const char* names[3] = {"v_mfma_f32_16x16x4_f32", "v_mfma_f32_16x16x16_bf16", "v_mfma_f32_16x16x32_bf16"};
    for (int waves = 1; waves <= 4; waves *= 4)
        for (int mode = 0; mode < 3; ++mode) {
            long long h[256];
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double m = 0; for (int i = 0; i < 256; ++i) m += (double)h[i]; m /= 256;
            printf("%-28s waves/WG %d: %.2f cycles per instruction per wave (8 independent accumulators)\n", names[mode], waves, m / (iters * 8.0));
        }
    return 0;
}
