// mfma_probe -- verifies the lane->element maps of v_mfma_i32_32x32x32_i8 / 16x16x64_i8 with exact
// integer data (guide: "check the map with exact integer data before relying on it").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); exit(1); } } while (0)

// assumed: lane l holds A[row l&31][k = 16*(l>>5) + j] and B[k = 16*(l>>5) + j][col l&31], j = 0..15;
// C[reg] = C[row (reg&3) + 8*(reg>>2) + 4*(l>>5)][col l&31]
__global__ void k32(const int8_t * A, const int8_t * B, int * C) { // A [32][32] row-major (row, k); B [32 k][32 col]
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    int8_t a[16], b[16];
    for (int j = 0; j < 16; j++) { a[j] = A[r * 32 + 16 * h + j]; b[j] = B[(16 * h + j) * 32 + r]; }
    v4i av, bv; __builtin_memcpy(&av, a, 16); __builtin_memcpy(&bv, b, 16);
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, c, 0, 0, 0);
    for (int reg = 0; reg < 16; reg++) C[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = c[reg];
}
// assumed: lane l holds A[row l&15][k = 16*(l>>4) + j], B[k = 16*(l>>4)+j][col l&15]; C[reg] = C[row 4*(l>>4)+reg][col l&15]
__global__ void k16(const int8_t * A, const int8_t * B, int * C) { // A [16][64], B [64][16]
    const int l = threadIdx.x, r = l & 15, h = l >> 4;
    int8_t a[16], b[16];
    for (int j = 0; j < 16; j++) { a[j] = A[r * 64 + 16 * h + j]; b[j] = B[(16 * h + j) * 16 + r]; }
    v4i av, bv; __builtin_memcpy(&av, a, 16); __builtin_memcpy(&bv, b, 16);
    v4i c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, c, 0, 0, 0);
    for (int reg = 0; reg < 4; reg++) C[(4 * h + reg) * 16 + r] = c[reg];
}
int main() {
    int8_t hA[2048], hB[2048]; int hC[1024], ref[1024];
    srand(7);
    for (int i = 0; i < 2048; i++) { hA[i] = rand() % 255 - 127; hB[i] = rand() % 255 - 127; }
    int8_t * dA, * dB; int * dC;
    CK(hipMalloc(&dA, 2048)); CK(hipMalloc(&dB, 2048)); CK(hipMalloc(&dC, 4096));
    CK(hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice));
    k32<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { int s = 0; for (int k = 0; k < 32; k++) s += hA[i * 32 + k] * hB[k * 32 + j]; ref[i * 32 + j] = s; if (s != hC[i * 32 + j]) bad++; }
    printf("mfma_i32_32x32x32_i8: %d mismatches of 1024\n", bad);
    k16<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost));
    bad = 0;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { int s = 0; for (int k = 0; k < 64; k++) s += hA[i * 64 + k] * hB[k * 16 + j]; if (s != hC[i * 16 + j]) bad++; }
    printf("mfma_i32_16x16x64_i8: %d mismatches of 256\n", bad);
    return 0;
}
