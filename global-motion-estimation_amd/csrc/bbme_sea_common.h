// Shared pieces of the successive-elimination exhaustive kernels (bbme_sea.hip: MAE,
// bbme_sea_mse.hip: MSE): launch descriptor, LDS layout, window staging, 8x8 box sums.
#pragma once
#include <stdlib.h>

#include "gme_internal.h"

namespace sea {

struct SeaDev {
    const uint8_t* prev;
    const uint8_t* cur;
    long long plane_stride;
    int pairs, H, W, pitch, sw;
    int nbr, nbc, nb, wg_per_row, wg_per_pair;
    int pitch_dw, win_rows;
    int32_t* mf;
    int xq;                       // S8 quads (4 columns each) per window row
    const uint32_t* sqbox;        // MSE only: 16x16 box sums of squares of `cur`, [pairs][H][pitch]
    long long sqbox_stride;
#ifdef GME_SEA_STAMPS
    long long* stamps;            // diagnostic build only (tools/microbench/sea_phases.hip): 8 per wave
#endif
};

typedef uint64_t u64_a4 __attribute__((aligned(4)));

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, m, 64));
    return v;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m, 64);
    return v;
}

// LDS carve-up shared by both kernels (dword offsets); `best` holds NB 64-bit slots so the MSE
// kernel can keep 37-bit keys there (the MAE kernel uses the low dword of each).
struct Layout {
    int win, anchor, best, count, a2, s8, work, total;
};

__host__ __device__ inline Layout make_layout(int R, int nb, int win_rows, int pitch_dw, int xq)
{
    Layout l;
    l.win = 0;
    l.anchor = win_rows * pitch_dw;
    l.best = (l.anchor + nb * 64 + 1) & ~1;            // 8-byte aligned
    l.count = l.best + 2 * nb;
    l.a2 = l.count + 2;
    l.s8 = (l.a2 + nb + 1) & ~1;                       // 8-byte aligned, [16R+8][xq] u16x4
    l.work = l.s8 + 2 * (16 * R + 8) * xq;             // [nb*64*R] entries
    l.total = l.work + nb * 64 * R;
    return l;
}

// A: stage the common search window of the workgroup's blocks (coalesced dword loads; rows and
// columns outside the frame -> 0).  Thread -> one dword column and every rstep-th row, loads in
// batches of four so that their latencies overlap.
__device__ __forceinline__ void stage_window(const SeaDev& d, uint32_t* win, const uint8_t* cur, int bcol0, int r0)
{
    const int T = blockDim.x;
    const int gx0 = bcol0 * 16 - d.sw, gy0 = r0 - d.sw;
    const int rstep = T / d.pitch_dw;
    const int row0 = threadIdx.x / d.pitch_dw, dw = threadIdx.x - row0 * d.pitch_dw;
    const int gx = gx0 + 4 * dw;
    const bool colok = gx >= 0 && gx < d.pitch;
    if (row0 < rstep) {
        const uint8_t* src = cur + (long long)(gy0 + row0) * d.pitch + gx;
        const long long sstep = (long long)rstep * d.pitch;
        uint32_t* dst = win + row0 * d.pitch_dw + dw;
        const int dstep = rstep * d.pitch_dw;
        for (int row = row0; row < d.win_rows; row += 4 * rstep, src += 4 * sstep, dst += 4 * dstep) {
            uint32_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int gy = gy0 + row + u * rstep;
                v[u] = 0;
                if (colok && row + u * rstep < d.win_rows && gy >= 0 && gy < d.H) v[u] = *(const uint32_t*)(src + u * sstep);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (row + u * rstep < d.win_rows) dst[u * dstep] = v[u];
        }
    }
}

// A': 8x8 box sums of the staged window.  Thread (column quad sq, row chunk ch) walks CH+7 window
// rows: per row two QSADs against a zero reference give the four horizontal 8-byte sums
// r8(row, 4sq .. 4sq+3) (packed u16); the vertical 8-row sum slides with a ring of 8 rows in
// registers: S8(y) = S8(y-1) + r8(y+7) - r8(y-1).  s8[y][sq] = packed S8(y, 4sq .. 4sq+3).
template <int R>
__device__ __forceinline__ void box_sums8(const SeaDev& d, const uint32_t* win, uint64_t* s8)
{
    typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
    constexpr int CH = 2 * R + 1;                      // 8 chunks cover 16R + 8 rows
    const int XQ = d.xq;
    for (int it = threadIdx.x; it < 8 * XQ; it += blockDim.x) {
        const int ch = it / XQ, sq = it - ch * XQ;
        const uint32_t* p = win + (ch * CH) * d.pitch_dw + sq;
        u16x4 ring[8], sum = { 0, 0, 0, 0 };
#pragma unroll
        for (int r = 0; r < CH + 7; ++r) {
            const uint64_t w0 = *(const u64_a4*)(p + r * d.pitch_dw), w1 = *(const u64_a4*)(p + r * d.pitch_dw + 1);
            const u16x4 h = __builtin_bit_cast(u16x4, __builtin_amdgcn_qsad_pk_u16_u8(
                                w1, 0u, __builtin_amdgcn_qsad_pk_u16_u8(w0, 0u, (uint64_t)0)));
            if (r >= 8) sum -= ring[r & 7];
            sum += h;
            ring[r & 7] = h;
            if (r >= 7) s8[(ch * CH + r - 7) * XQ + sq] = __builtin_bit_cast(uint64_t, sum);
        }
    }
}

// quadrant sums of the anchor held one dword per lane (lane = row * 4 + dword): xor 1 pairs the
// two dwords of a half row, xor 4/8/16 sums the 8 rows of a half -> lanes 0, 2, 32, 34.
__device__ __forceinline__ void anchor_quadrants(uint32_t mine, uint32_t* a01, uint32_t* a23)
{
    uint32_t s = __builtin_amdgcn_sad_u8(mine, 0u, 0u);
    s += (uint32_t)__shfl_xor((int)s, 1, 64);
    s += (uint32_t)__shfl_xor((int)s, 4, 64);
    s += (uint32_t)__shfl_xor((int)s, 8, 64);
    s += (uint32_t)__shfl_xor((int)s, 16, 64);
    *a01 = (uint32_t)__builtin_amdgcn_readlane((int)s, 0) | ((uint32_t)__builtin_amdgcn_readlane((int)s, 2) << 16);
    *a23 = (uint32_t)__builtin_amdgcn_readlane((int)s, 32) | ((uint32_t)__builtin_amdgcn_readlane((int)s, 34) << 16);
}

inline int pick_pitch(int need, int R)
{
    int best_p = need, best_c = 1 << 30;
    for (int p = need; p < need + 33; ++p) {
        int conflicts = 0, seen[32];
        for (int i = 0; i < 32; ++i) seen[i] = 0;
        for (int prow = 0; prow < 8; ++prow)
            for (int q = 0; q < 4; ++q) conflicts += seen[((prow * R) * p + q * R) & 31]++;
        if (conflicts < best_c) { best_c = conflicts; best_p = p; }
    }
    return best_p;
}

// Host: waves (= macroblocks) per workgroup and the LDS it needs.  More blocks share more of the
// staged window and of the box-sum pass, but LDS per workgroup grows; pick the count that keeps
// most waves resident per CU (160 KiB LDS, 32 waves), discounted by the idle waves of a ragged
// last workgroup.  Returns false if nothing fits.
inline bool plan(int R, int nbc, SeaDev* d, size_t* lds_bytes)
{
    auto bytes_for = [&](int nb_, int* pitch_out, int* xq_out) {
        const int xq = (4 * R + 4 * (nb_ - 1) + 2) | 1;                 // S8 quads per row, odd pitch
        const int need_dw = (nb_ - 1) * 4 + 3 * R + (R - 1) + 5;        // base + k + 4 pairs of two dwords
        const int pitch = pick_pitch(need_dw > xq + 2 ? need_dw : xq + 2, R);
        if (pitch_out) *pitch_out = pitch;
        if (xq_out) *xq_out = xq;
        return (size_t)make_layout(R, nb_, 16 * R + 15, pitch, xq).total * 4;
    };
    int nb = 0;
    double best_score = -1.0;
    for (int cand = 1; cand <= 16 && cand <= nbc; ++cand) {
        const size_t bytes = bytes_for(cand, nullptr, nullptr);
        if (bytes > 160 * 1024) break;
        const int wgs = (int)((160 * 1024) / (bytes + 1024));           // allocation granularity slack
        const int waves = wgs * cand > 32 ? 32 : wgs * cand;
        const int per_row = (nbc + cand - 1) / cand;
        // measured (NB sweep at 720x480): wave counts that spread evenly over the 4 SIMDs (8, 16) beat
        // 15 by 13 % although 3 of 48 wave slots per block row idle -> small bonus for multiples of 4
        const double score = waves * ((double)nbc / (per_row * cand)) + cand * 1e-3 + (cand % 4 == 0 ? 0.5 : 0.0);
        if (score > best_score) { best_score = score; nb = cand; }
    }
    if (const char* e = getenv("GME_SEA_NB")) nb = atoi(e) < 1 ? 1 : (atoi(e) > 16 ? 16 : atoi(e));
    if (nb > nbc) nb = nbc;
    if (nb < 1) return false;
    d->nb = nb;
    d->wg_per_row = (nbc + nb - 1) / nb;
    d->win_rows = 16 * R + 15;
    *lds_bytes = bytes_for(nb, &d->pitch_dw, &d->xq);
    return *lds_bytes <= 160 * 1024;
}

}  // namespace sea
