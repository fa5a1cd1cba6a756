// Layout probe of ds_read_b64_tr_b8 (gfx950): which LDS bytes does lane l receive?
// Hypothesis H (by analogy with ds_read_b64_tr_b16, where lane 4q+p of a 16-lane group supplies the address of (row q, columns
// 4p..4p+3) and lane i receives column i of rows 0..3): lane 2q+p of a 16-lane group supplies the address of (row q, byte columns
// 8p..8p+7) of an 8-row x 16-column byte block, and lane i receives column i, rows 0..7, one per byte in row order.
//   hipcc --offload-arch=gfx950 -O2 -o tr_b8_layout tr_b8_layout.hip && ./tr_b8_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v2i __attribute__((ext_vector_type(2)));
constexpr int ROWB = 128;

__global__ void probe(unsigned char* out_row, unsigned char* out_col) {
    __shared__ __attribute__((aligned(16))) unsigned char img[64 * ROWB];
    const int l = threadIdx.x;
    for (int which = 0; which < 2; ++which) {
        for (int i = l; i < 64 * ROWB; i += 64) img[i] = which ? (unsigned char)(i % ROWB) : (unsigned char)(i / ROWB);
        __syncthreads();
        const int grp = l >> 4, i16 = l & 15;
        const int q = i16 >> 1, p = i16 & 1;
        const unsigned char* addr = img + (grp * 8 + q) * ROWB + 32 + p * 8;       // group g: rows 8g..8g+7, columns 32..47
        v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3)))*)(uintptr_t)addr);
        unsigned char* o = (which ? out_col : out_row) + l * 8;
        for (int j = 0; j < 8; ++j) o[j] = (unsigned char)((j < 4 ? r[0] >> (8 * j) : r[1] >> (8 * (j - 4))) & 0xFF);
        __syncthreads();
    }
}

int main() {
    unsigned char *dr, *dc;
    hipMalloc(&dr, 512); hipMalloc(&dc, 512);
    probe<<<1, 64>>>(dr, dc);
    std::vector<unsigned char> r(512), c(512);
    hipMemcpy(r.data(), dr, 512, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 8; ++j) {
            printf(" (%2d,%2d)", r[l * 8 + j], c[l * 8 + j]);
            const int want_row = (l >> 4) * 8 + j, want_col = 32 + (l & 15);
            if (r[l * 8 + j] != want_row || c[l * 8 + j] != want_col) ++bad;
        }
        printf("\n");
    }
    printf("hypothesis H: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return 0;
}
