// rank_bw_gen1.h -- diagnostic only: the first-generation rank-bw update kernel (register-staged LDS,
// row map fetched per k-tile), kept so that tools/rank_bw_bench.hip can check the product kernel
// (gpu_matrix_inversion_amd/csrc/mi32_rank_bw.h) against it bit for bit and A/B their speed.
#pragma once
namespace mi32 {
#ifndef MI32_BW_BK
#define MI32_BW_BK 16
#endif
#ifndef MI32_BW_WPS
#define MI32_BW_WPS 3
#endif
// ---- the rank-bw update, pipelined --------------------------------------------------
// Same arithmetic as gj_rank_update_kernel<128,128,32,false> (one k-ascending MFMA chain per output
// element, so results are bit-identical), restructured for the matrix pipe:
//  * register-staged double buffering: the global loads of k-tile t+1 are issued before the 64 MFMAs of
//    k-tile t and written to the other LDS buffer after them -> one barrier per k-tile instead of two,
//    and the loads' latency hides under the MFMAs;
//  * XCD-aware tile order: workgroups that share an XCD (blockIdx % 8) cover a compact (T/2) x (T/4)
//    sub-grid of tiles, so that XCD's 4 MiB L2 holds the A and B panels its tiles re-read.
template <int BK>
__device__ __forceinline__ void rank_bw_tile(const float *__restrict__ src_all, float *__restrict__ dst_all,
                                             const float *__restrict__ g_all, size_t gstride, int np, int ld,
                                             size_t mstride, int c0, int kdim, const int *__restrict__ map_all,
                                             int copy_panel, float *__restrict__ pt_out_all, size_t tstride,
                                             int pt_col, int pt_w, int skip_lo, int skip_hi, int b, int rt, int ct,
                                             float (&s_a)[2][BK * (128 + (BK == 32 ? 1 : 2))],
                                             float (&s_b)[2][BK * (128 + 4)], int (&s_map)[128])
{
    constexpr int BM = 128, BN = 128;
    constexpr int NQ = BK / 8;  // float4 per thread per operand tile
    constexpr int LDA = BM + (BK == 32 ? 1 : 2), LDB = BN + 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int row0 = rt * BM, col0 = ct * BN;
    const float *src = src_all + (size_t)b * mstride;
    float *dst = dst_all + (size_t)b * mstride;
    const float *g = g_all + (size_t)b * gstride;
    const int *map = map_all + (size_t)b * np;
    float *pt_out = pt_out_all + (size_t)b * tstride;

    if (col0 >= skip_lo && col0 < skip_hi) return;
    if (col0 >= c0 && col0 + BN <= c0 + kdim) {  // tile inside the panel: those columns are G itself
        if (copy_panel) {
            for (int idx = tid; idx < BM * (BN / 4); idx += 256) {
                const int rr = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
                *reinterpret_cast<float4 *>(dst + (size_t)(row0 + rr) * ld + col0 + c4) =
                    *reinterpret_cast<const float4 *>(g + (size_t)(row0 + rr) * ld + col0 + c4);
            }
        }
        return;
    }

    // staging: A tile 128 rows x BK k (float4 idx: row = idx / (BK/4), k4 = idx % (BK/4)),
    //          B tile BK k x 128 columns (k = idx / 32, c4 = idx % 32); idx = tid + 256 q, q < BK/8
    typedef float f4v __attribute__((ext_vector_type(4)));  // native vector: HIP's float4 struct in an array goes to scratch
    f4v ra[NQ], rb[NQ];
#define MI32_LOAD_TILES(KT)                                                                                        \
    _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                                               \
        const int idx = tid + q * 256;                                                                             \
        ra[q] = *reinterpret_cast<const f4v *>(g + (size_t)(row0 + idx / (BK / 4)) * ld + c0 + (KT) +              \
                                               (idx % (BK / 4)) * 4);                                              \
        rb[q] = *reinterpret_cast<const f4v *>(src + (size_t)map[c0 + (KT) + (idx >> 5)] * ld + col0 +             \
                                                  (idx & 31) * 4);                                                 \
    }
#define MI32_STORE_TILES(BUF)                                                                  \
    _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                           \
        const int idx = tid + q * 256;                                                         \
        float *pa = &s_a[BUF][((idx % (BK / 4)) * 4) * LDA + idx / (BK / 4)];                  \
        pa[0] = ra[q][0]; pa[LDA] = ra[q][1]; pa[2 * LDA] = ra[q][2]; pa[3 * LDA] = ra[q][3];  \
        *reinterpret_cast<f4v *>(&s_b[BUF][(idx >> 5) * LDB + (idx & 31) * 4]) = rb[q];        \
    }

    MI32_LOAD_TILES(0)
    if (tid < BM) s_map[tid] = map[row0 + tid];
    __syncthreads();

    // The MFMA chain starts from zero and the old values C (row-mapped; 0 for the rows of the block
    // itself) are added AFTER the k-loop: sum of products first (k ascending), then + C.  That keeps only
    // the 64 accumulators live across the loop (4 workgroups per CU) and is the more accurate order
    // (residual 9e-5 instead of 3e-4 at N = 4096: the rounding error of the sum no longer scales with
    // |C|).  oracle/gj_oracle.c's blocked mirror uses the same order.
    // (Tried: checkerboarding "C first" / "C last" over the tiles to de-synchronise the memory-bound and
    // the MFMA phases of co-resident workgroups -- no gain, and it makes the rounding depend on where a
    // row is stored, which breaks the exact invariance inv(P A) == inv(A) P^T.)
    float16v acc[2][2];
    const int lcol = lane & 31;
    const int lhalf = lane >> 5;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) acc[tm][tn][reg] = 0.0f;
    MI32_STORE_TILES(0)
    __syncthreads();

    const int nk = kdim / BK;
    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        const int ktn = (t + 1 < nk) ? (t + 1) * BK : t * BK;  // last iteration: harmless re-load, keeps the loop branch-free
        MI32_LOAD_TILES(ktn)
        // fragments of k-pair kk+2 are read from LDS BEFORE the four MFMAs of k-pair kk are issued, so the
        // LDS latency hides under 256 MFMA cycles (hipcc otherwise emits read -> lgkmcnt(0) -> MFMAs per pair)
        float af[2], bf[2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) af[tm] = s_a[buf][lhalf * LDA + wr * 64 + tm * 32 + lcol];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) bf[tn] = s_b[buf][lhalf * LDB + wc * 64 + tn * 32 + lcol];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float afn[2] = {0.f, 0.f}, bfn[2] = {0.f, 0.f};
            if (kk + 2 < BK) {
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) afn[tm] = s_a[buf][(kk + 2 + lhalf) * LDA + wr * 64 + tm * 32 + lcol];
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) bfn[tn] = s_b[buf][(kk + 2 + lhalf) * LDB + wc * 64 + tn * 32 + lcol];
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch reads ABOVE this pair's MFMAs
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm], bf[tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 2; ++q) { af[q] = afn[q]; bf[q] = bfn[q]; }
        }
        MI32_STORE_TILES(buf ^ 1)
        __syncthreads();
    }

#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int col = col0 + wc * 64 + tn * 32 + lcol;
            const bool exp = (col >= pt_col && col < pt_col + pt_w);  // next sub-panel's column
            // all 16 old values of this 32x32 sub-tile first (independent loads, in flight together), then
            // add + store: interleaved, every load would have to wait for the store before it (may-alias).
            // (Requesting the next sub-tile's values before storing this one needs 16 more registers and
            // drops the occupancy from 4 to 3 workgroups per CU: measured slower, 118 vs 101 us.)
            float cv[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int lr = wr * 64 + tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lhalf;
                const int grow = row0 + lr;
                const bool in_block = (grow >= c0 && grow < c0 + kdim);
                cv[reg] = in_block ? 0.0f : src[(size_t)s_map[lr] * ld + col];
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int grow = row0 + wr * 64 + tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lhalf;
                const float v = acc[tm][tn][reg] + cv[reg];
                dst[(size_t)grow * ld + col] = v;
                if (exp) pt_out[(size_t)(col - pt_col) * np + grow] = v;
            }
        }
#undef MI32_LOAD_TILES
#undef MI32_STORE_TILES
}

// XCD-aware tile order: workgroups that share an XCD (linear id % 8) cover a compact (T/2) x (T/4)
// sub-grid of tiles, so that XCD's 4 MiB L2 holds the A and B panels its tiles re-read.
__device__ __forceinline__ void rank_bw_tile_of(int id, int T, int &rt, int &ct)
{
    if ((T & 7) == 0) {
        const int xcd = id & 7, idx = id >> 3;
        const int tr = T / 2, tc = T / 4;
        rt = (xcd >> 2) * tr + idx / tc;
        ct = (xcd & 3) * tc + idx % tc;
    } else {
        rt = id / T;
        ct = id % T;
    }
}

template <int BK, int WPS>
__global__ __launch_bounds__(256, WPS) void gj_rank_bw_update_kernel(const float *__restrict__ src_all,
                                                                     float *__restrict__ dst_all,
                                                                     const float *__restrict__ g_all, size_t gstride,
                                                                     int np, int ld, size_t mstride, int c0, int kdim,
                                                                     const int *__restrict__ map_all, int copy_panel,
                                                                     float *__restrict__ pt_out_all, size_t tstride,
                                                                     int pt_col, int pt_w, int skip_lo, int skip_hi)
{
    __shared__ float s_a[2][BK * (128 + (BK == 32 ? 1 : 2))];
    __shared__ __attribute__((aligned(16))) float s_b[2][BK * (128 + 4)];
    __shared__ int s_map[128];
    int rt, ct;
    rank_bw_tile_of(blockIdx.x, np / 128, rt, ct);
    rank_bw_tile<BK>(src_all, dst_all, g_all, gstride, np, ld, mstride, c0, kdim, map_all, copy_panel, pt_out_all,
                     tstride, pt_col, pt_w, skip_lo, skip_hi, blockIdx.y, rt, ct, s_a, s_b, s_map);
}

// Persistent, residency-limited flavour for the look-ahead half: gridDim.x workgroups walk all the tiles.
// It is launched with enough dynamic LDS that only ONE workgroup fits on a CU and with fewer workgroups
// than CUs, so a known number of CUs stays entirely free for the critical-path kernels of the main
// stream (the panel kernel needs a whole CU); stream priorities cannot give that guarantee and a CU
// mask serialises the queues.
template <int BK>
__global__ __launch_bounds__(256, 1) void gj_rank_bw_update_persistent_kernel(
    const float *__restrict__ src_all, float *__restrict__ dst_all, const float *__restrict__ g_all, size_t gstride,
    int np, int ld, size_t mstride, int c0, int kdim, const int *__restrict__ map_all, int copy_panel,
    float *__restrict__ pt_out_all, size_t tstride, int pt_col, int pt_w, int skip_lo, int skip_hi)
{
    __shared__ float s_a[2][BK * (128 + (BK == 32 ? 1 : 2))];
    __shared__ __attribute__((aligned(16))) float s_b[2][BK * (128 + 4)];
    __shared__ int s_map[128];
    const int T = np / 128;
    for (int id = blockIdx.x; id < T * T; id += gridDim.x) {
        int rt, ct;
        rank_bw_tile_of(id, T, rt, ct);
        rank_bw_tile<BK>(src_all, dst_all, g_all, gstride, np, ld, mstride, c0, kdim, map_all, copy_panel, pt_out_all,
                         tstride, pt_col, pt_w, skip_lo, skip_hi, blockIdx.y, rt, ct, s_a, s_b, s_map);
        __syncthreads();  // the next tile re-uses the LDS buffers
    }
}

}  // namespace mi32
