/* spatial_clip_hip.h -- C ABI of the MI355X (gfx950) kernels behind the Spatial-CLIP training step.
 * (header grows with the kernels; see bottom of file for the full list)
 */
#ifndef SPATIAL_CLIP_HIP_H
#define SPATIAL_CLIP_HIP_H
#ifdef __cplusplus
extern "C" {
#endif
enum { SC_GEMM_NT = 0, SC_GEMM_TN = 1 };
enum { SC_EPI_BF16 = 0, SC_EPI_BF16_BIAS = 1, SC_EPI_F32_BIAS_RES = 2, SC_EPI_GELU_PAIR = 3,
       SC_EPI_BF16_DGELU = 4, SC_EPI_F32 = 5 };
const char* sc_last_error(void);
int sc_gemm_bf16(int mode, int epi, const void* A, int lda, const void* B, int ldb, int M, int N, int K,
                 void* C, int ldc, void* C2, int ldc2, const float* bias, const float* res, int ldres,
                 const void* aux, int ldaux, int splitk, float* slabs, void* stream);
long long sc_gemm_slab_floats(int M, int N, int K, int splitk);
#ifdef __cplusplus
}
#endif
#endif
