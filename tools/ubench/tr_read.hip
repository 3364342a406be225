// semantics of ds_read_b64_tr_b16 on gfx950: every lane supplies the address of 4 contiguous 16-bit elements; prints what lane l receives
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/tr_read.hip -o tr_read
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ short smem[256];
    for (int i = threadIdx.x; i < 256; i += 64) smem[i] = (short)i;
    __syncthreads();
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + 4 * threadIdx.x));
    *reinterpret_cast<s16x4*>(out + 4 * threadIdx.x) = v;
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            const int G = l >> 4, c = l & 15, expect = 4 * (16 * G + 4 * j + (c >> 2)) + (c & 3);
            printf(" %3d", h[4 * l + j]);
            bad += h[4 * l + j] != expect;
        }
        printf("\n");
    }
    printf("mismatches against [lane c, elem j] <- element (c & 3) of lane 16 G + 4 j + (c >> 2): %d\n", bad);
    return 0;
}
