// Operand layout + rate probe of v_mfma_scale_f32_16x16x128_f8f6f4 with OCP e4m3 operands (gfx950).
// Hypothesis H (checked against an integer reference): lane l supplies row (l & 15) of A / column (l & 15) of B and the
// 32 CONSECUTIVE k values 32 * (l >> 4) .. +31, one per byte of its 8-VGPR operand, in byte order; scales E8M0 = 127
// (1.0) in every byte; C/D layout as for the bf16 16x16 shapes (col = lane & 15, row = 4 * (lane >> 4) + reg).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_fp8_layout.hip -o /tmp/fp8probe && /tmp/fp8probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static unsigned char enc(int v) {           // exact e4m3fn code of an integer in [-7, 7]
    static const unsigned char pos[8] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E};
    return v < 0 ? (unsigned char)(0x80 | pos[-v]) : pos[v];
}

__global__ void probe(const unsigned char* A, const unsigned char* B, float* D) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    i32x8 a, b;
    const int* pa = reinterpret_cast<const int*>(A + r * 128 + g * 32);
    const int* pb = reinterpret_cast<const int*>(B + r * 128 + g * 32);
    for (int j = 0; j < 8; ++j) { a[j] = pa[j]; b[j] = pb[j]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int j = 0; j < 4; ++j) D[(g * 4 + j) * 16 + r] = c[j];
}

__global__ void rate(float* out, int iters) {
    i32x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = 0x38383838 + threadIdx.x; b[j] = 0x40404040 ^ (threadIdx.x << 3); }
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[i], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    std::vector<unsigned char> A(16 * 128), B(16 * 128);
    std::vector<int> Ai(16 * 128), Bi(16 * 128);
    srand(1);
    for (int i = 0; i < 16 * 128; ++i) { Ai[i] = rand() % 15 - 7; Bi[i] = rand() % 15 - 7; A[i] = enc(Ai[i]); B[i] = enc(Bi[i]); }
    unsigned char *dA, *dB; float* dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD);
    std::vector<float> D(256);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        long ref = 0;
        for (int k = 0; k < 128; ++k) ref += (long)Ai[m * 128 + k] * Bi[n * 128 + k];
        if ((long)D[m * 16 + n] != ref) { if (bad < 5) printf("mismatch D[%d][%d] = %g, ref %ld\n", m, n, D[m * 16 + n], ref); ++bad; }
    }
    printf("layout hypothesis H (D[m][n] = sum_k A[m][k] B[n][k], first operand rows on the MFMA row index): %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    // if H fails with operands swapped, D holds the transpose: report that too
    int badT = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        long ref = 0;
        for (int k = 0; k < 128; ++k) ref += (long)Ai[n * 128 + k] * Bi[m * 128 + k];
        if ((long)D[m * 16 + n] != ref) ++badT;
    }
    printf("transposed reading: %s\n", badT ? "no" : "OK");
    float* dO; hipMalloc(&dO, 256 * 4 * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate<<<1024, 256>>>(dO, 100);
    hipEventRecord(e0); rate<<<1024, 256>>>(dO, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 16 * 16 * 128 * 8.0 * iters * 1024 * 4;
    printf("rate: %.1f TFLOP/s (fp8 e4m3, scaled 16x16x128, 4 waves/CU x 4 blocks/CU)\n", flops / ms / 1e9);
    return bad ? 1 : 0;
}
