// Exhaustive block matching, bs = 16, MAE -- the headline kernel (BASELINE config 2).
//
// Replaces the Python loops of bbme.py:105-179 (one compute_dfd call per candidate,
// bbme.py:41-64) for block_size 16 and pnorm 0.
//
// Mapping (one wavefront = one macroblock, as BASELINE.json's north_star asks):
//   * a workgroup of NB waves owns NB horizontally adjacent macroblocks and stages
//     their common search window ((2sw+31) rows) of `cur` into LDS with coalesced
//     dword loads;
//   * each wave keeps its 16x16 anchor block of `prev` in 64 SGPRs (scalar loads):
//     the anchor is wave-uniform, so it costs no vector registers and no LDS reads;
//   * lane (pr = lane/4, q = lane%4) owns R candidate rows x 4R candidate columns
//     (NC = 16R candidates per axis); the inner instruction is v_qsad_pk_u16_u8, which
//     yields four 4-byte SADs at four consecutive byte offsets of a 64-bit window --
//     the sliding window a motion search needs -- accumulating four packed u16 sums
//     (a 16x16 SAD is at most 65280, so u16 never overflows);
//   * the winner is the minimum of (sad << 13 | scan_index) keys, scan_index =
//     col_idx * NC + row_idx, i.e. the reference's "first strict minimum, column
//     offset outer loop" rule (bbme.py:146-149,171), reduced across the wave.
//
// XCD-aware launch: workgroups are dealt round-robin over the 8 XCDs, so workgroup b
// handles pair (b/8/WPP)*8 + b%8 -- all blocks of one frame pair run on one XCD and
// the pair's two frames are pulled into that XCD's L2 once.
#include "gme_internal.h"

namespace {

struct FastDev {
    const uint8_t* prev;
    const uint8_t* cur;
    long long plane_stride;
    int pairs, H, W, pitch, sw;
    int nbr, nbc, nb, wg_per_row, wg_per_pair;
    int pitch_dw, win_rows;
    int32_t* mf;
};

typedef uint64_t u64_a4 __attribute__((aligned(4)));

template <int R>
__global__ void k_exh_qsad16(FastDev d)
{
    extern __shared__ uint32_t win[];                 // [win_rows][pitch_dw]
    constexpr int NW = R + 3;                          // 64-bit window pairs per lane and row
    const int b = blockIdx.x;
    const int pair = (b / 8 / d.wg_per_pair) * 8 + (b & 7);
    if (pair >= d.pairs) return;                       // whole workgroup leaves together
    const int wg = (b >> 3) % d.wg_per_pair;
    const int brow = wg / d.wg_per_row;
    const int bcol0 = (wg - brow * d.wg_per_row) * d.nb;
    const int r0 = brow * 16;
    const int NC = 2 * d.sw + 16;
    const uint8_t* cur = d.cur + (long long)pair * d.plane_stride;

    // ---- stage the search window (coalesced dword loads; out-of-frame -> 0) ----
    // thread -> one dword column and every `rstep`-th row: no div/mod inside the loop
    {
        const int gx0 = bcol0 * 16 - d.sw;             // multiple of 4 (sw % 4 == 0)
        const int gy0 = r0 - d.sw;
        const int rstep = blockDim.x / d.pitch_dw;
        const int row0 = threadIdx.x / d.pitch_dw, dw = threadIdx.x - row0 * d.pitch_dw;
        const int gx = gx0 + 4 * dw;
        const bool colok = gx >= 0 && gx < d.pitch;
        if (row0 < rstep) {
            const uint8_t* src = cur + (long long)(gy0 + row0) * d.pitch + gx;
            const long long sstep = (long long)rstep * d.pitch;
            uint32_t* dst = win + row0 * d.pitch_dw + dw;
            const int dstep = rstep * d.pitch_dw;
            for (int row = row0; row < d.win_rows; row += rstep, src += sstep, dst += dstep) {
                const int gy = gy0 + row;
                uint32_t v = 0;
                if (colok && gy >= 0 && gy < d.H) v = *(const uint32_t*)src;
                *dst = v;
            }
        }
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bcol = bcol0 + wave;
    if (bcol >= d.nbc) return;                         // ragged last workgroup (after the barrier)
    const int c0 = bcol * 16;
    const int lane = threadIdx.x & 63;
    const int prow = lane >> 2, q = lane & 3;

    // ---- anchor block: 64 wave-uniform dwords (scalar loads -> SGPRs) ----
    const uint32_t* anchor = (const uint32_t*)(d.prev + (long long)pair * d.plane_stride +
                                               (long long)r0 * d.pitch + c0);
    const int apitch = d.pitch >> 2;
    uint32_t A[16][4];
#pragma unroll
    for (int a = 0; a < 16; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) A[a][j] = anchor[a * apitch + j];

    // ---- sliding SAD over this lane's R x 4R candidates ----
    uint64_t acc[R][R];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int k = 0; k < R; ++k) acc[i][k] = 0;

    const uint32_t* lrow = win + (prow * R) * d.pitch_dw + wave * 4 + q * R;
#pragma unroll
    for (int t = 0; t < R + 15; ++t) {
        uint64_t w[NW];
#pragma unroll
        for (int s = 0; s < NW; ++s) w[s] = *(const u64_a4*)(lrow + t * d.pitch_dw + s);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int a = t - i;                       // anchor row matched by window row t
            if (a < 0 || a > 15) continue;
#pragma unroll
            for (int k = 0; k < R; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][k] = __builtin_amdgcn_qsad_pk_u16_u8(w[k + j], A[a][j], acc[i][k]);
        }
    }

    // ---- pick the first minimum in the reference's scan order ----
    // Lane-local keys: sad << 16 | local, local = (4k+e)*R + i grows with the scan order
    // (column index outer, row index inner) inside a lane; one op builds a key from a packed
    // accumulator half: (dword << 16) | local for the low u16, (dword & 0xFFFF0000) | local
    // for the high one.
    const int lo_r = max(0, d.sw - r0), hi_r = min(NC - 1, d.H - 16 - r0 + d.sw);
    const int lo_c = max(0, d.sw - c0), hi_c = min(NC - 1, d.W - 16 - c0 + d.sw);
    const bool interior = __builtin_amdgcn_readfirstlane(lo_r == 0 && lo_c == 0 && hi_r == 16 * R - 1 && hi_c == 16 * R - 1);
    uint32_t lbest = 0xFFFFFFFFu;
    if (interior) {
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int k = 0; k < R; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t dwv = (uint32_t)(acc[i][k] >> (32 * (e >> 1)));
                    const uint32_t local = (uint32_t)((4 * k + e) * R + i);
                    const uint32_t key = (e & 1) ? ((dwv & 0xFFFF0000u) | local) : ((dwv << 16) | local);
                    lbest = min(lbest, key);
                }
    } else {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int ri = prow * R + i;
            const bool rok = ri >= lo_r && ri <= hi_r;
#pragma unroll
            for (int k = 0; k < R; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ci = q * 4 * R + 4 * k + e;
                    const uint32_t dwv = (uint32_t)(acc[i][k] >> (32 * (e >> 1)));
                    const uint32_t local = (uint32_t)((4 * k + e) * R + i);
                    const uint32_t key = (e & 1) ? ((dwv & 0xFFFF0000u) | local) : ((dwv << 16) | local);
                    if (rok && ci >= lo_c && ci <= hi_c) lbest = min(lbest, key);
                }
        }
    }
    uint32_t best = 0xFFFFFFFFu;
    if (lbest != 0xFFFFFFFFu) {
        const int local = lbest & 0xFFFF;
        const int ce = local / R, i = local - ce * R;
        best = ((lbest >> 16) << 13) | (uint32_t)((q * 4 * R + ce) * NC + prow * R + i);
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, m, 64));
    if (lane == 0) {
        const int idx = best & 0x1FFF;
        const int ci = idx / NC, ri = idx - ci * NC;
        int32_t* o = d.mf + (((long long)pair * d.nbr + brow) * d.nbc + bcol) * 2;
        o[0] = ci - d.sw;
        o[1] = ri - d.sw;
    }
}

// LDS row pitch (in dwords) that keeps the per-row ds_read2_b32 of a half-wave on
// distinct banks: lanes (prow 0..7, q 0..3) read dword (prow*R + t)*pitch + q*R + s.
int pick_pitch_dw(int need, int R)
{
    int best_p = need, best_c = 1 << 30;
    for (int p = need; p < need + 33; ++p) {
        int conflicts = 0;
        int seen[32];
        for (int i = 0; i < 32; ++i) seen[i] = 0;
        for (int prow = 0; prow < 8; ++prow)
            for (int q = 0; q < 4; ++q) {
                const int bank = ((prow * R) * p + q * R) & 31;
                conflicts += seen[bank]++;
            }
        if (conflicts < best_c) { best_c = conflicts; best_p = p; }
    }
    return best_p;
}

}  // namespace

int launch_bbme_fast(gme_ctx* ctx, const BbmeJob& job, bool* handled)
{
    *handled = false;
    if (job.procedure != GME_SEARCH_EXHAUSTIVE || job.bs != 16 || job.pnorm != GME_NORM_MAE) return GME_OK;
    if (job.sw < 0 || job.sw % 4 != 0) return GME_OK;
    const int NC = 2 * job.sw + 16;
    const int R = (NC + 15) / 16;
    if (R < 1 || R > 5) return GME_OK;                 // NC*NC scan index must also fit 13 bits
    if (NC * NC > 8192) return GME_OK;
    const int nbr = job.H / 16, nbc = job.W / 16;
    if (nbr == 0 || nbc == 0) return GME_OK;

    FastDev d;
    d.prev = job.prev; d.cur = job.cur; d.plane_stride = job.plane_stride;
    d.pairs = job.pairs; d.H = job.H; d.W = job.W; d.pitch = job.pitch; d.sw = job.sw;
    d.nbr = nbr; d.nbc = nbc; d.mf = job.mf;
    // waves per workgroup: prefer an exact divisor of the block-row length
    int nb = 4;
    for (int cand : {5, 4, 6, 3}) if (nbc % cand == 0) { nb = cand; break; }
    if (nbc < nb) nb = nbc;
    d.nb = nb;
    d.wg_per_row = (nbc + nb - 1) / nb;
    d.wg_per_pair = d.wg_per_row * nbr;
    d.win_rows = 16 * R + 15;                          // rows a lane may touch: prow*R + t, t < R+15
    // bytes a lane may touch in a row: wave*16 + q*4R + 4*(R+3) + 8  (last 64-bit pair)
    const int need_dw = (nb - 1) * 4 + 3 * R + (R + 2) + 2;
    d.pitch_dw = pick_pitch_dw(need_dw, R);
    const size_t lds = (size_t)d.win_rows * d.pitch_dw * 4;
    const long long groups = (long long)((job.pairs + 7) / 8) * 8 * d.wg_per_pair;
    GME_REQUIRE(groups < (1ll << 31), GME_ERR_ARG, "too many workgroups in one launch");
    const dim3 grid((unsigned)groups), block(64 * nb);
    switch (R) {
    case 1: hipLaunchKernelGGL(k_exh_qsad16<1>, grid, block, lds, ctx->stream, d); break;
    case 2: hipLaunchKernelGGL(k_exh_qsad16<2>, grid, block, lds, ctx->stream, d); break;
    case 3: hipLaunchKernelGGL(k_exh_qsad16<3>, grid, block, lds, ctx->stream, d); break;
    case 4: hipLaunchKernelGGL(k_exh_qsad16<4>, grid, block, lds, ctx->stream, d); break;
    default: hipLaunchKernelGGL(k_exh_qsad16<5>, grid, block, lds, ctx->stream, d); break;
    }
    GME_HIP_TRY(hipGetLastError());
    *handled = true;
    return GME_OK;
}
