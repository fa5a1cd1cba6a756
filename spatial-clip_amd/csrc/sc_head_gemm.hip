// Exact-fp32 MFMA GEMM of the contrastive head (gfx950): the cosine-similarity logits z = f . all_f^T and the four
// gradient products behind them run on the matrix cores with v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate:
// bit-for-bit an fmaf chain, 1/16 of the bf16 MFMA rate = the f32 vector peak, with the VALU left free).
//   reference op sites: src/open_clip/loss.py:116-118 (logits_per_image / logits_per_text),
//                       src/models/components/losses.py:78-81 (z_i_t / z_t_i) and their autograd.
// One kernel serves every operand layout of the head through element strides:
//   C[m][n] (+)= sum_k A[m*sam + k*sak] * B[n*sbn + k*sbk]
//   forward   z      = f[B,D]    . all_f[G,D]^T     A k-contiguous, B k-contiguous
//   backward  d f    = dz[B,G]   . all_f[G,D]       A k-contiguous, B n-contiguous
//             d all  = dz[B,G]^T . f[B,D]           A m-contiguous, B n-contiguous
// Up to SC_SGEMM_MAX_GROUP independent problems share one launch (a flat tile list): the two directions of the forward
// and the four backward products fill the chip together instead of running as six under-filled launches.
//
// Tile: TM x TN x 32, 256 threads = 2 x 2 waves, each wave (TM/2) x (TN/2) as 16x16 MFMA fragments.  Operand tiles are
// staged in LDS in their *natural* layout (coalesced 16-byte global loads, 16-byte LDS stores, next tile prefetched
// into registers under the MFMAs):
//   k-contiguous operand: [row][32 k + 4 pad]; a lane reads 16 bytes = k {4g..4g+3} of its row (g = lane>>4) and
//                          feeds component j to MFMA j of the 16-k chunk, so MFMA j sums k = 4g + j over the 4 groups;
//   n-contiguous operand: [k][TN + 4 pad]; a lane reads the single float (k = 4g + j, its column): the same k
//                          assignment, so any layout pairs with any other.  The pad makes both reads (near) conflict-free.
// Shapes are arbitrary: tiles on the M / N / K edge or with unaligned strides take a guarded scalar staging path.
#include "sc_common.h"
#include "sc_kernels.h"

namespace {

constexpr int HK = 32;            // K depth of an LDS tile
constexpr int MAXG = SC_SGEMM_MAX_GROUP;

struct Problem {
    const float* A; long long sam, sak;
    const float* B; long long sbn, sbk;
    float* C; long long ldc;
    int M, N, K, accumulate, tiles_n, tile_end;
};
struct Group { Problem p[MAXG]; int n; };

template <int TILE>
SC_DEVICE void stage_load(const float* __restrict__ base, long long srow, long long sk, int row0, int k0, int rows,
                          int K, bool kc, bool fast, int t, f32x4 (&reg)[TILE / 32]) {
    // TILE rows x 32 k  = TILE*8 float4 -> TILE/32 per thread
#pragma unroll
    for (int p = 0; p < TILE / 32; ++p) {
        const int idx = p * 256 + t;
        int r, k;
        if (kc) { r = idx >> 3; k = (idx & 7) * 4; }                 // [row][k]: 8 float4 per row
        else { k = idx / (TILE / 4); r = (idx % (TILE / 4)) * 4; }   // [k][row]: TILE/4 float4 per k
        if (fast) {
            reg[p] = *reinterpret_cast<const f32x4*>(base + (long long)(row0 + r) * srow + (long long)(k0 + k) * sk);
        } else {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int rr = row0 + r + (kc ? 0 : e), kk = k0 + k + (kc ? e : 0);
                if (rr < rows && kk < K) v[e] = base[(long long)rr * srow + (long long)kk * sk];
            }
            reg[p] = v;
        }
    }
}

template <int TILE>
SC_DEVICE void stage_store(float* __restrict__ lds, bool kc, int t, const f32x4 (&reg)[TILE / 32]) {
#pragma unroll
    for (int p = 0; p < TILE / 32; ++p) {
        const int idx = p * 256 + t;
        if (kc) *reinterpret_cast<f32x4*>(lds + (idx >> 3) * (HK + 4) + (idx & 7) * 4) = reg[p];
        else *reinterpret_cast<f32x4*>(lds + (idx / (TILE / 4)) * (TILE + 4) + (idx % (TILE / 4)) * 4) = reg[p];
    }
}

template <int TM, int TN>
__global__ __launch_bounds__(256) void head_gemm_kernel(const Group grp) {
    constexpr int FM = TM / 32, FN = TN / 32;          // 16x16 fragments per wave in M / N
    constexpr int LA = TM * (HK + 4) > HK * (TM + 4) ? TM * (HK + 4) : HK * (TM + 4);
    constexpr int LB = TN * (HK + 4) > HK * (TN + 4) ? TN * (HK + 4) : HK * (TN + 4);
    __shared__ __attribute__((aligned(16))) float lds[LA + LB];
    float* As = lds;
    float* Bs = lds + LA;

    // which problem / tile is this block's (block-uniform scalar search over <= MAXG entries)
    int pi = 0, tile = blockIdx.x;
    {
        int begin = 0;
#pragma unroll
        for (int i = 0; i < MAXG; ++i) {
            if (i < grp.n && (int)blockIdx.x >= grp.p[i].tile_end) { pi = i + 1; begin = grp.p[i].tile_end; }
        }
        tile -= begin;
    }
    const Problem& P = grp.p[pi];
    const int m0 = (tile / P.tiles_n) * TM, n0 = (tile % P.tiles_n) * TN;
    const int M = P.M, N = P.N, K = P.K;
    const bool akc = P.sak == 1, bkc = P.sbk == 1;
    const long long a_row = P.sam, a_k = P.sak, b_row = P.sbn, b_k = P.sbk;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (wave >> 1) * (TM / 2), wn = (wave & 1) * (TN / 2);
    const int l15 = lane & 15, g = lane >> 4;

    // 16-byte global loads need: the contiguous dimension's stride 1 (given), the other stride % 4 == 0, an aligned
    // base and a tile fully inside the matrix (K edge handled per tile below)
    const bool a_al = ((reinterpret_cast<uintptr_t>(P.A) & 15) == 0) && (akc ? (a_row & 3) == 0 : ((a_k & 3) == 0 && a_row == 1));
    const bool b_al = ((reinterpret_cast<uintptr_t>(P.B) & 15) == 0) && (bkc ? (b_row & 3) == 0 : ((b_k & 3) == 0 && b_row == 1));
    const bool a_in = m0 + TM <= M, b_in = n0 + TN <= N;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 ra[TM / 32], rb[TN / 32];
    const int nk = (K + HK - 1) / HK;
    {
        const bool kin = HK <= K;
        stage_load<TM>(P.A, a_row, a_k, m0, 0, M, K, akc, a_al && a_in && kin, t, ra);
        stage_load<TN>(P.B, b_row, b_k, n0, 0, N, K, bkc, b_al && b_in && kin, t, rb);
    }
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                                   // previous tile's fragment reads are done
        stage_store<TM>(As, akc, t, ra);
        stage_store<TN>(Bs, bkc, t, rb);
        __syncthreads();
        if (kt + 1 < nk) {                                 // prefetch the next tile under this tile's MFMAs
            const int k0 = (kt + 1) * HK;
            const bool kin = k0 + HK <= K;
            stage_load<TM>(P.A, a_row, a_k, m0, k0, M, K, akc, a_al && a_in && kin, t, ra);
            stage_load<TN>(P.B, b_row, b_k, n0, k0, N, K, bkc, b_al && b_in && kin, t, rb);
        }
#pragma unroll
        for (int ks = 0; ks < HK / 16; ++ks) {
            float af[FM][4], bfr[FN][4];
            if (akc) {
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(As + (wm + i * 16 + l15) * (HK + 4) + ks * 16 + 4 * g);
                    af[i][0] = v[0]; af[i][1] = v[1]; af[i][2] = v[2]; af[i][3] = v[3];
                }
            } else {
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) af[i][j] = As[(ks * 16 + 4 * g + j) * (TM + 4) + wm + i * 16 + l15];
            }
            if (bkc) {
#pragma unroll
                for (int i = 0; i < FN; ++i) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(Bs + (wn + i * 16 + l15) * (HK + 4) + ks * 16 + 4 * g);
                    bfr[i][0] = v[0]; bfr[i][1] = v[1]; bfr[i][2] = v[2]; bfr[i][3] = v[3];
                }
            } else {
#pragma unroll
                for (int i = 0; i < FN; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) bfr[i][j] = Bs[(ks * 16 + 4 * g + j) * (TN + 4) + wn + i * 16 + l15];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int n = 0; n < FN; ++n)
                        acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][j], bfr[n][j], acc[i][n], 0, 0, 0);
        }
    }
    // D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int n = 0; n < FN; ++n) {
            const int gn = n0 + wn + n * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm + i * 16 + g * 4 + r;
                if (gm < M && gn < N) {
                    float* c = P.C + (long long)gm * P.ldc + gn;
                    *c = P.accumulate ? *c + acc[i][n][r] : acc[i][n][r];
                }
            }
        }
}

// [B, D] features | two int64 id vectors -> one fp32 row of D (+4) floats: the send buffer of the feature all-gather
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ feat, long long ldf,
                                                        const long long* __restrict__ ids_a,
                                                        const long long* __restrict__ ids_b, float* __restrict__ out,
                                                        long long ldo, int B, int D) {
    const int row = blockIdx.x;
    const float* f = feat + (long long)row * ldf;
    float* o = out + (long long)row * ldo;
    for (int c = threadIdx.x; c < D; c += blockDim.x) o[c] = f[c];
    if (ids_a && threadIdx.x == 0) {
        long long* oi = reinterpret_cast<long long*>(o + D);      // D even and ldo even: 8-byte aligned
        oi[0] = ids_a[row];
        oi[1] = ids_b[row];
    }
}

}  // namespace

extern "C" int sc_sgemm_f32_grouped(const sc_sgemm_desc* descs, int n, void* stream) {
    SC_CHECK(descs != nullptr && n >= 1 && n <= MAXG, "sc_sgemm_f32_grouped: 1..%d problems, got %d", MAXG, n);
    Group g;
    g.n = n;
    // one tile size for the group: 128x128 when every problem is large in both output dimensions (fewer operand
    // re-reads), 64x64 otherwise (the head's M = local batch is small: more blocks fill the chip)
    long long tiles128 = 0;
    bool big = true;
    for (int i = 0; i < n; ++i) {
        const sc_sgemm_desc& d = descs[i];
        SC_CHECK(d.M > 0 && d.N > 0 && d.K > 0, "sc_sgemm_f32_grouped: empty problem %d", i);
        SC_CHECK(d.A && d.B && d.C, "sc_sgemm_f32_grouped: null operand in problem %d", i);
        SC_CHECK(d.sam == 1 || d.sak == 1, "sc_sgemm_f32_grouped: A of problem %d has no unit stride", i);
        SC_CHECK(d.sbn == 1 || d.sbk == 1, "sc_sgemm_f32_grouped: B of problem %d has no unit stride", i);
        tiles128 += (long long)((d.M + 127) / 128) * ((d.N + 127) / 128);
        big = big && d.M >= 128 && d.N >= 128;
    }
    const bool use128 = big && tiles128 >= 1024;
    const int T = use128 ? 128 : 64;
    long long total = 0;
    for (int i = 0; i < n; ++i) {
        const sc_sgemm_desc& d = descs[i];
        Problem& p = g.p[i];
        p.A = d.A; p.sam = d.sam; p.sak = d.sak;
        p.B = d.B; p.sbn = d.sbn; p.sbk = d.sbk;
        p.C = d.C; p.ldc = d.ldc;
        p.M = d.M; p.N = d.N; p.K = d.K; p.accumulate = d.accumulate;
        p.tiles_n = (d.N + T - 1) / T;
        total += (long long)((d.M + T - 1) / T) * p.tiles_n;
        SC_CHECK(total < (1ll << 30), "sc_sgemm_f32_grouped: too many tiles");
        p.tile_end = (int)total;
    }
    for (int i = n; i < MAXG; ++i) { g.p[i] = g.p[n - 1]; }
    hipStream_t st = (hipStream_t)stream;
    if (use128) head_gemm_kernel<128, 128><<<(unsigned)total, 256, 0, st>>>(g);
    else head_gemm_kernel<64, 64><<<(unsigned)total, 256, 0, st>>>(g);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_sgemm_f32(const float* A, long long sam, long long sak, const float* B, long long sbn, long long sbk,
                            float* C, long long ldc, int M, int N, int K, int accumulate, void* stream) {
    sc_sgemm_desc d;
    d.A = A; d.sam = sam; d.sak = sak; d.B = B; d.sbn = sbn; d.sbk = sbk; d.C = C; d.ldc = ldc;
    d.M = M; d.N = N; d.K = K; d.accumulate = accumulate;
    return sc_sgemm_f32_grouped(&d, 1, stream);
}

extern "C" int sc_pack_rows(const float* feat, long long ldf, const long long* ids_a, const long long* ids_b,
                            float* out, long long ldo, int B, int D, void* stream) {
    SC_CHECK(B > 0 && D > 0 && ldf >= D, "sc_pack_rows: bad shape B=%d D=%d", B, D);
    SC_CHECK((ids_a == nullptr) == (ids_b == nullptr), "sc_pack_rows: pass both id vectors or neither");
    SC_CHECK(ldo >= D + (ids_a ? 4 : 0), "sc_pack_rows: output rows too short");
    SC_CHECK(!ids_a || ((D & 1) == 0 && (ldo & 1) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0),
             "sc_pack_rows: ids need D, ldo even and an 8-byte aligned output");
    pack_rows_kernel<<<B, 256, 0, (hipStream_t)stream>>>(feat, ldf, ids_a, ids_b, out, ldo, B, D);
    SC_LAUNCH_CHECK();
    return 0;
}
