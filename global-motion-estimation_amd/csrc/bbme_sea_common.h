// Shared pieces of the successive-elimination exhaustive kernels (bbme_sea.hip: MAE,
// bbme_sea_mse.hip: MSE): launch descriptor, LDS layout, tile shapes, window staging, 8x8 box sums,
// the persistent tile driver.
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include "gme_internal.h"

namespace sea {

// weight of window sharing in the tile-shape score (plan()); tuned by same-box sweeps
#ifndef SEA_SHARE_WEIGHT
#define SEA_SHARE_WEIGHT 0.5
#endif

struct SeaDev {
    const uint8_t* prev;
    const uint8_t* cur;
    long long plane_stride;
    int pairs, H, W, pitch, sw;
    int nbr, nbc;
    int tr, tc, nb;               // a tile is tr x tc macroblocks, one wave each: nb = tr * tc waves per workgroup
    int wg_per_row, tile_rows, wg_per_pair;      // tiles per block row / tile rows / tiles per pair
    uint32_t magic_tc;            // wave / tc (div_small)
    int pitch_dw, win_rows;       // staged window: 16R + 15 + 16 (tr - 1) rows of pitch_dw dwords
    int s8_rows;                  // rows of the box-sum table: 16R + 8 + 16 (tr - 1)
    int rstep;                    // staging: window rows covered by one sweep of the workgroup (T / pitch_dw)
    uint32_t magic_pitch;         // n / pitch_dw == (n * magic_pitch) >> 20 for n < 4096 (div_small)
    uint32_t magic_xq;            // same for n / xq
    unsigned long long magic_wpp; // persistent kernel: n / wg_per_pair == (n * magic_wpp) >> 40 for n < 2^21
    unsigned long long magic_wpr; // same for n / wg_per_row
    uint32_t* status;             // the context's status words (gme_internal.h: GME_STATUS_*):
                                  //   + GME_STATUS_TILECTR  persistent kernel, dynamic schedule: one tile counter per XCD, 16 words apart
                                  //   + GME_STATUS_STATS    per XCD, 16 words apart: [0] patches evaluated exactly (phase E), [1] tiles handed to
                                  //                         the brute-force redo kernel, [2] patches phase D listed (what one
                                  //                         round would have evaluated; == [0] without ordered evaluation)
                                  //   + GME_STATUS_REDO     [0] length of redo_list
    int dynamic;                  // persistent kernel: draw tiles from the per-XCD counters (else a static stride)
    // Hostile content (nothing correlates: a scene cut, noise): the bound prunes little and phase E's one-patch-
    // per-lane evaluation costs more than evaluating everything in the regular layout of k_exh_qsad16 / k_exh_dot16.
    // A tile whose list is longer than redo_threshold skips phases E and F and goes to redo_list instead
    // (entry = tile number inside its XCD's tiles << 3 | xcd); the redo kernel launched behind this one searches it.
    uint32_t* redo_list;          // or null: no fallback
    int redo_threshold;
    // Ordered evaluation (round 4, phase C2 of the kernels): a block whose first upper bound leaves more than `quota`
    // patches scores those with the smallest bounds inside its own wave first (threshold by `bisect` halvings of [smallest
    // bound, UB], at most `quota` <= 16 pass) and lists the rest against the upper bound that leaves.  On real content the
    // smallest bound does not name the best candidate, but the best one sits among the small bounds: phase D then sees
    // (almost) the block's true minimum (tools/ub_study.py).  quota <= 0: off.  `engage`: survivors from which a block
    // counts as crowded (>= quota).
    int quota, bisect, engage;
    int32_t* mf;
    int xq;                       // S8 quads (4 columns each) per window row
    const uint32_t* sqbox;        // MSE only: 16x16 box sums of squares of `cur`, [pairs][H][pitch]
    long long sqbox_stride;
#ifdef GME_SEA_STAMPS
    long long* stamps;            // diagnostic build only (tools/microbench/sea_phases.hip): 8 per wave
#endif
};

typedef uint64_t u64_a4 __attribute__((aligned(4)));

// wave_min_u32 / wave_sum_u32 (DPP reductions): gme_internal.h
#define SEA_DPP(v, ctrl) GME_DPP(v, ctrl)

// n / dv for n < 4096 and dv < 256 without the 20-odd instructions of an emulated division:
// magic = 2^20 / dv + 1 overshoots the reciprocal by < 2^-20, so the product is off by < 2^-8 < 1 / dv.
constexpr uint32_t div_magic(int dv) { return (1u << 20) / (uint32_t)dv + 1u; }
// same idea for n < 2^21, dv < 2^18: (n * magic40) >> 40, error < 2^21 / 2^40 < 1 / dv
inline unsigned long long div_magic40(int dv) { return (1ull << 40) / (unsigned long long)dv + 1ull; }
__device__ __forceinline__ int div_small(int n, uint32_t magic) { return (int)(__umul24((uint32_t)n, magic) >> 20); }

// Workgroup -> (pair, tile row, first block column).  Grid = (8 * wg_per_row, tile_rows, ceil(pairs / 8)):
// workgroups go to the 8 XCDs round robin in x-fastest order, so the low 3 bits of blockIdx.x pick
// the pair inside a group of 8 and all tiles of one pair land on one XCD (its L2 holds the pair).
__device__ __forceinline__ bool locate(const SeaDev& d, int* pair, int* trow, int* bcol0)
{
    *pair = (int)blockIdx.z * 8 + (int)(blockIdx.x & 7);
    *trow = (int)blockIdx.y;
    *bcol0 = (int)(blockIdx.x >> 3) * d.tc;
    return *pair < d.pairs;
}

inline bool grid_for(const SeaDev& d, dim3* grid)
{
    const long long gz = ((long long)d.pairs + 7) / 8;
    if (gz > 65535 || d.tile_rows > 65535) return false;
    *grid = dim3((unsigned)(8 * d.wg_per_row), (unsigned)d.tile_rows, (unsigned)gz);
    return true;
}

// Anchors sit 68 dwords apart, not 64: phase E lanes serving different blocks read the same anchor
// element of "their" block at once, and a 64-dword stride would put all of those in one LDS bank.
constexpr int ANCHOR_STRIDE = 68;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Layout {
    int win, anchor, best, count, a2, prev, own, s8, work, total;
};

__host__ __device__ constexpr Layout make_layout(int R, int nb, int win_rows, int pitch_dw, int xq, int s8_rows)
{
    Layout l{};
    l.win = 0;
    l.anchor = (win_rows * pitch_dw + 3) & ~3;         // 16-byte aligned: anchor rows are read as b128
    l.best = (l.anchor + nb * ANCHOR_STRIDE + 1) & ~1; // 8-byte aligned
    l.count = l.best + 2 * nb;                         // [0] list length, [1] next tile, [3] streak of hostile tiles (persistent_tiles),
                                                       // [4] patches phase C2 took off the list (scored or pruned), [5] patches C2 scored,
                                                       // [7] MSE: length of the second list (level 2 of the bound)
    l.a2 = l.count + 8;
    l.prev = l.a2 + nb;                                // [nb] scan index each wave's block of the previous tile ended with (third probe)
    l.own = l.prev + nb;                               // [nb][16] phase C2: the patches a crowded block's wave scores itself
    l.s8 = (l.own + 16 * nb + 1) & ~1;                 // 8-byte aligned, [s8_rows][xq] u16x4
    l.work = l.s8 + 2 * s8_rows * xq;                  // [nb*64*R] entries
    l.total = l.work + nb * 64 * R;
    return l;
}

__device__ __forceinline__ Layout layout_of(const SeaDev& d, int R)
{
    return make_layout(R, d.nb, d.win_rows, d.pitch_dw, d.xq, d.s8_rows);
}

// The macroblock a wave owns inside its tile (waves are numbered row-major over the tile).
struct WaveBlock {
    int wr, wc;                   // position inside the tile
    int brow, bcol;               // block coordinates in the frame
    bool ok;                      // inside the frame's block grid (ragged last tile of a row / column)
};
__device__ __forceinline__ WaveBlock wave_block(const SeaDev& d, int trow, int bcol0, int wave)
{
    WaveBlock b;
    b.wr = div_small(wave, d.magic_tc);
    b.wc = wave - b.wr * d.tc;
    b.brow = trow * d.tr + b.wr;
    b.bcol = bcol0 + b.wc;
    b.ok = b.bcol < d.nbc && b.brow < d.nbr;
    return b;
}

// A: stage the common search window of the workgroup's blocks (coalesced dword loads; rows and
// columns outside the frame -> 0).  Thread -> one dword column and every rstep-th row, loads in
// batches of four so that their latencies overlap.
__device__ __forceinline__ void stage_window(const SeaDev& d, uint32_t* win, const uint8_t* cur, int bcol0, int r0)
{
    const int gx0 = bcol0 * 16 - d.sw, gy0 = r0 - d.sw;
    const int rstep = d.rstep;
    const int row0 = div_small((int)threadIdx.x, d.magic_pitch), dw = (int)threadIdx.x - row0 * d.pitch_dw;
    const int gx = gx0 + 4 * dw;
    const bool colok = gx >= 0 && gx < d.pitch;
    if (row0 < rstep) {
        const uint8_t* src = cur + (long long)(gy0 + row0) * d.pitch + gx;
        const long long sstep = (long long)rstep * d.pitch;
        uint32_t* dst = win + threadIdx.x;                     // == row0 * pitch_dw + dw
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
// CH = s8_rows / 8 is a template parameter so that the ring indices and the warm-up are resolved
// at compile time (a run-time row count cost 5 % of the whole search in guards).
template <int CH, int NCH = 8>
__device__ __forceinline__ void box_sums8_ch(const SeaDev& d, const uint32_t* win, uint64_t* s8, int tid)
{
    typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
    const int XQ = d.xq;
    for (int it = tid; it < NCH * XQ; it += blockDim.x) {
        const int ch = div_small(it, d.magic_xq), sq = it - ch * XQ;
        int pi = (ch * CH) * d.pitch_dw + sq, oi = (ch * CH) * XQ + sq;     // running offsets: adds, no r * pitch multiplies
        u16x4 ring[8], sum = { 0, 0, 0, 0 };
#pragma unroll
        for (int r = 0; r < CH + 7; ++r) {
            const uint64_t w0 = *(const u64_a4*)(win + pi), w1 = *(const u64_a4*)(win + pi + 1);
            pi += d.pitch_dw;
            asm volatile("" : "+v"(pi));
            const u16x4 h = __builtin_bit_cast(u16x4, __builtin_amdgcn_qsad_pk_u16_u8(
                                w1, 0u, __builtin_amdgcn_qsad_pk_u16_u8(w0, 0u, (uint64_t)0)));
            if (r >= 8) sum -= ring[r & 7];
            sum += h;
            ring[r & 7] = h;
            if (r >= 7) {
                s8[oi] = __builtin_bit_cast(uint64_t, sum);
                oi += XQ;
                asm volatile("" : "+v"(oi));
            }
        }
    }
}

// s8_rows = 16R + 8 + 16 (tr - 1) is a multiple of 8 for every tile height: 8 chunks of 2R + 2 tr - 1 rows
// (SEA_BOX_CHUNKS = 4: four chunks of twice the rows for two-row tiles -- 7 warm-up rows per chunk are 44 % of a 9-row
// chunk's work and 28 % of an 18-row one's, but half as many lanes share it; A/B in DESIGN.md)
#ifndef SEA_BOX_CHUNKS
#define SEA_BOX_CHUNKS 8
#endif
template <int R>
__device__ __forceinline__ void box_sums8(const SeaDev& d, const uint32_t* win, uint64_t* s8, int tid)
{
    if (d.tr == 1) box_sums8_ch<2 * R + 1>(d, win, s8, tid);
    else if (d.tr == 2) {
        constexpr int ROWS = 16 * R + 24;                  // s8_rows of a two-row tile
        if (SEA_BOX_CHUNKS != 8 && ROWS % SEA_BOX_CHUNKS == 0) box_sums8_ch<ROWS / SEA_BOX_CHUNKS, SEA_BOX_CHUNKS>(d, win, s8, tid);
        else box_sums8_ch<2 * R + 3>(d, win, s8, tid);
    }
    else box_sums8_ch<2 * R + 7>(d, win, s8, tid);
}

// quadrant sums of the anchor held one dword per lane (lane = row * 4 + dword): the quad swap pairs
// the two dwords of a half row, row_ror 4 and 8 add the four rows inside a 16-lane DPP row, the two
// DPP rows of each half are added on the scalar side.
__device__ __forceinline__ void anchor_quadrants(uint32_t mine, uint32_t* a01, uint32_t* a23)
{
    uint32_t s = __builtin_amdgcn_sad_u8(mine, 0u, 0u);
    s += SEA_DPP(s, 0xB1);                                 // quad_perm [1,0,3,2]
    s += SEA_DPP(s, 0x124);                                // row_ror 4
    s += SEA_DPP(s, 0x128);                                // row_ror 8
    const uint32_t q0 = (uint32_t)__builtin_amdgcn_readlane((int)s, 0) + (uint32_t)__builtin_amdgcn_readlane((int)s, 16);
    const uint32_t q1 = (uint32_t)__builtin_amdgcn_readlane((int)s, 2) + (uint32_t)__builtin_amdgcn_readlane((int)s, 18);
    const uint32_t q2 = (uint32_t)__builtin_amdgcn_readlane((int)s, 32) + (uint32_t)__builtin_amdgcn_readlane((int)s, 48);
    const uint32_t q3 = (uint32_t)__builtin_amdgcn_readlane((int)s, 34) + (uint32_t)__builtin_amdgcn_readlane((int)s, 50);
    *a01 = q0 | (q1 << 16);
    *a23 = q2 | (q3 << 16);
}

// LDS pitches from bank models of the reads that dominate (MI355X_MICROARCH.md, LDS: a wave's 8-byte reads are
// served 32 lanes at a time over 64 four-byte banks, 4-byte reads over 32; lanes conflict only on different
// addresses of one bank).
//
// Box-sum table pitch XQ (8-byte quads per row): phase B's lane (prow, q) reads quads (prow * R + i) * XQ + q * R + k,
// so the 32 lanes of a half wave sit prow * R * XQ + q * R quads apart.  An odd pitch (the first choice) left a
// two-way conflict on every one of those reads at R = 3 and a three-way one at R = 5; the pitch is now the
// smallest one >= need with the fewest conflicts (28 and 44 quads: none).
constexpr int box_conflicts(int xq, int R)
{
    int worst = 0, hits[64] = {};
    for (int b = 0; b < 64; ++b) hits[b] = 0;
    for (int prow = 0; prow < 8; ++prow)
        for (int q = 0; q < 4; ++q) {
            const int dw = 2 * ((prow * R) * xq + q * R);
            ++hits[dw & 63]; ++hits[(dw + 1) & 63];             // distinct lanes read distinct quads here
        }
    for (int b = 0; b < 64; ++b) worst = hits[b] > worst ? hits[b] : worst;
    return worst;
}
constexpr int pick_xq(int need, int R)
{
    int best = need, best_c = 1 << 30;
    for (int x = need; x < need + 12; ++x) {
        const int c = box_conflicts(x, R);
        if (c < best_c) { best_c = c; best = x; }
    }
    return best;
}

// Window pitch (dwords).  Phase E, four lanes per patch: the lanes of a quad read the same columns of window rows
// 4 apart, i.e. 4 * pitch dwords apart -- with pitch % 8 == 4 (the first choice, made for the brute-force kernel's
// access pattern, which this kernel does not have) lanes 0/2 and 1/3 of every quad hit the same bank.  Phase C's two
// probes (lane = 4 * row + dword, rows `pitch` apart) are rare and weigh little.
constexpr int pick_pitch(int need, int R)
{
    int best_p = need, best_c = 1 << 30;
    for (int p = need; p < need + 16; ++p) {
        int quad = 0, seen[32] = {};
        for (int i = 0; i < 32; ++i) seen[i] = 0;
        for (int sub = 0; sub < 4; ++sub) { quad += seen[(4 * sub * p) & 31]; ++seen[(4 * sub * p) & 31]; quad += seen[(4 * sub * p + 1) & 31]; }
        int probe = 0;
        for (int i = 0; i < 32; ++i) seen[i] = 0;
        for (int row = 0; row < 8; ++row)
            for (int dw = 0; dw < 4; ++dw) probe += seen[(row * p + dw) & 31]++;
        // a wider row also costs staging loads per thread (rows per sweep = threads / pitch) and LDS
        const int c = 64 * quad + probe + 2 * (p - need);
        if (c < best_c) { best_c = c; best_p = p; }
    }
    (void)R;
    return best_p;
}

// Host: tile shape (tr x tc macroblocks = waves per workgroup) and the LDS it needs.  A larger tile
// shares more of the staged window and of the box-sum pass between its blocks (window bytes per
// block: 1193 for 1 x 16, 864 for 2 x 8 at sw = 16), but LDS per workgroup grows.  The score is the
// number of waves a CU keeps resident (160 KiB LDS, 32 waves), discounted by the idle waves of
// ragged last tiles and by SIMD imbalance, with a bonus for the sharing.  GME_SEA_TILE = "TRxTC"
// overrides it (A/B runs).  Returns false if nothing fits.
// LDS geometry of a tile shape: depends on (R, tr, tc) only, so a kernel instantiated for one shape
// (fix_geometry below) gets every stride, row count and division constant at compile time.
struct Shape { int tr, tc, pitch, xq; size_t bytes; };
constexpr Shape shape_of(int R, int tr, int tc)
{
    Shape s{};
    s.tr = tr; s.tc = tc;
    s.xq = pick_xq(4 * R + 4 * (tc - 1) + 2, R);                    // S8 quads per row
    const int need_dw = (tc - 1) * 4 + 3 * R + (R - 1) + 5;         // base + k + 4 pairs of two dwords
    s.pitch = pick_pitch(need_dw > s.xq + 2 ? need_dw : s.xq + 2, R);
    s.bytes = (size_t)make_layout(R, tr * tc, 16 * R + 15 + 16 * (tr - 1), s.pitch, s.xq, 16 * R + 8 + 16 * (tr - 1)).total * 4;
    return s;
}

// Kernels instantiated for one tile shape (GEO = tr * 16 + tc, 0 = any shape at run time) overwrite the
// geometry fields of their by-value launch descriptor with constants: the compiler then folds every use
// (strides become immediates, the multiply-shift divisions constants, ~15 SGPRs are never loaded).
// The host launches such an instance only when plan() chose exactly that shape (geometry_matches).
template <int R, int GEO>
__device__ __forceinline__ void fix_geometry(SeaDev& d)
{
    if constexpr (GEO != 0) {
        constexpr int TR = GEO / 16, TC = GEO % 16;
        constexpr Shape s = shape_of(R, TR, TC);
        d.tr = TR; d.tc = TC; d.nb = TR * TC;
        d.magic_tc = div_magic(TC);
        d.win_rows = 16 * R + 15 + 16 * (TR - 1);
        d.s8_rows = 16 * R + 8 + 16 * (TR - 1);
        d.pitch_dw = s.pitch; d.xq = s.xq;
        d.rstep = 64 * TR * TC / s.pitch;
        d.magic_pitch = div_magic(s.pitch);
        d.magic_xq = div_magic(s.xq);
        d.sw = 8 * R - 8;                                  // the full window of the size class: NC = 16 R, also a constant now
    }
}

inline bool geometry_matches(const SeaDev& d, int R, int geo)
{
    if (geo == 0) return true;
    const Shape s = shape_of(R, geo / 16, geo % 16);
    return d.tr == geo / 16 && d.tc == geo % 16 && d.pitch_dw == s.pitch && d.xq == s.xq && d.sw == 8 * R - 8;
}

inline bool plan(int R, int nbr, int nbc, int sw, SeaDev* d, size_t* lds_bytes)
{
    auto shape = [&](int tr, int tc) { return shape_of(R, tr, tc); };
    Shape best = shape(1, 1);
    double best_score = -1.0;
    int force_tr = 0, force_tc = 0;
    if (const char* e = getenv("GME_SEA_TILE")) sscanf(e, "%dx%d", &force_tr, &force_tc);
    if (const char* e = getenv("GME_SEA_NB")) { force_tr = 1; force_tc = atoi(e); }
    // resident waves per CU: 32 slots at <= 64 VGPRs (R <= 3); the R >= 4 kernels are held to 80 VGPRs -> 24
    const int wave_cap = R <= 3 ? 32 : 24;
    for (int pass = 0; pass < 2 && best_score < 0; ++pass)      // pass 0: SIMD-balanced wave counts only
        for (int tr = 1; tr <= 4; tr *= 2)
            for (int tc = 1; tc * tr <= 16; ++tc) {
                if (force_tr > 0 && (tr != force_tr || tc != force_tc)) continue;
                if (force_tr == 0 && (tc > nbc || (tr > 1 && tr > nbr) || (pass == 0 && (tr * tc) % 4 != 0))) continue;
                const Shape s = shape(tr, tc);
                if (s.bytes > 160 * 1024 || s.pitch >= 256 || s.xq >= 256) continue;
                const int nb = tr * tc;
                int wgs = (int)((160 * 1024) / (s.bytes + 1024));       // allocation granularity slack
                if (wgs * nb > wave_cap) wgs = wave_cap / nb;
                if (wgs < 1) wgs = 1;
                const int waves = wgs * nb;
                const double cover = ((double)nbc / (((nbc + tc - 1) / tc) * tc)) * ((double)nbr / (((nbr + tr - 1) / tr) * tr));
                // A workgroup's waves are dealt round robin to the CU's 4 SIMDs, so a wave count that is not a
                // multiple of 4 leaves one SIMD with an extra wave for the whole search (measured: 16 vs 15
                // waves +13 % at 720x480 sw 16; 8 vs 9 +17 %, 8 vs 12 (VGPR-limited to one workgroup) +30 %
                // at 1080p sw 32): such shapes are only used when nothing else fits.
                const double simd_eff = (double)nb / (4 * ((nb + 3) / 4));
                // staged window bytes per block, relative to a block's own (16 + 2 sw + 15)^2 window
                const double area = (double)(16 * tr + 2 * sw + 15) * (16 * tc + 2 * sw + 15) / nb;
                const double own = (double)(2 * sw + 31) * (2 * sw + 31);
                const double share = 1.0 + SEA_SHARE_WEIGHT * (1.0 - area / own);
                // a lone workgroup per CU has nobody to cover its barriers (measured 2-5 % at 1080p sw 32); four
                // 8-wave workgroups interleave better than two 16-wave ones (2 x 4 vs 2 x 8 tiles: +4.5 % at sw 16)
                const double together = wgs == 1 ? 0.93 : 1.0 + 0.03 * ((wgs > 4 ? 4 : wgs) - 2);
                // four-row tiles measured 3 % (720x480: 4 x 2 vs 2 x 4) to 15 % (1080p sw 32: 4 x 3 vs 2 x 6) slower
                // than two-row tiles of the same size: their windows are tall, staging rows are short
                const double tall = tr == 4 ? 0.95 : 1.0;
                const double score = waves * cover * simd_eff * share * together * tall + nb * 1e-3;
                if (score > best_score) { best_score = score; best = s; }
            }
    if (best_score < 0) return false;
    d->tr = best.tr; d->tc = best.tc; d->nb = best.tr * best.tc;
    d->magic_tc = div_magic(d->tc);
    d->wg_per_row = (nbc + d->tc - 1) / d->tc;
    d->tile_rows = (nbr + d->tr - 1) / d->tr;
    d->wg_per_pair = d->wg_per_row * d->tile_rows;
    d->win_rows = 16 * R + 15 + 16 * (d->tr - 1);
    d->s8_rows = 16 * R + 8 + 16 * (d->tr - 1);
    d->pitch_dw = best.pitch; d->xq = best.xq;
    *lds_bytes = best.bytes;
    d->rstep = 64 * d->nb / d->pitch_dw;
    d->magic_pitch = div_magic(d->pitch_dw);
    d->magic_xq = div_magic(d->xq);
    d->magic_wpp = div_magic40(d->wg_per_pair);
    d->magic_wpr = div_magic40(d->wg_per_row);
    return d->rstep >= 1;
}

constexpr int SEA_DEFAULT_QUOTA = 16, SEA_DEFAULT_BISECT = 5, SEA_DEFAULT_ENGAGE = 24;     // tools/ub_study.py; same-box A/B in DESIGN.md
constexpr int REDO_BURST = 15;
constexpr double REDO_DEFAULT_FRAC = 0.75;     // break-even measured on noise content

__device__ __forceinline__ void push_redo(const SeaDev& d, int tile_in_xcd, int xcd)
{
    const uint32_t slot = atomicAdd(d.status + GME_STATUS_REDO, 1u);
    d.redo_list[slot] = ((uint32_t)tile_in_xcd << 3) | (uint32_t)xcd;
    atomicAdd(d.status + GME_STATUS_STATS + 16 * xcd + 1, 1u);
}

// Phase C2's choice of a crowded block's first patches (wave-uniform): keys below the returned limit are scored first.
// `below(lim)` counts the wave's patches whose key is below lim; lo / hi are the smallest bound and the UB (same unit,
// `shift` = position of the bound inside a key).  Invariant: at most `quota` keys lie below (lo + 1) << shift, or lo is
// the smallest bound itself (ties may exceed the quota: they all go first).
template <class Below>
__device__ __forceinline__ uint32_t first_round_limit(const SeaDev& d, uint32_t lo, uint32_t hi, int shift, Below below)
{
    if (d.bisect < 0) return (lo + ((hi - lo) >> -d.bisect) + 1) << shift;      // no search: the lowest 1 / 2^n of the range
    for (int s = 0; s < d.bisect && lo < hi; ++s) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((int)below((mid + 1) << shift) > d.quota) hi = mid; else lo = mid;
    }
    return (lo + 1) << shift;
}

// tile number (inside its XCD's tiles) of the one-tile kernels' workgroup: what persistent_tiles calls `t`
__device__ __forceinline__ int tile_number(const SeaDev& d, int pair, int trow, int bcol0)
{
    return (pair >> 3) * d.wg_per_pair + trow * d.wg_per_row + div_small(bcol0, d.magic_tc);
}

// Persistent form of a search kernel: G workgroups (as many as fit on the chip at once) walk the
// tiles of "their" XCD's pairs.  The window and anchor of the next tile are fetched into registers
// while the current one is searched, so the HBM/L2 latency of phase A overlaps phases A' .. F
// instead of idling the workgroup's waves.  Kern supplies prep() (anchor -> LDS, per-wave anchor
// statistics) and phases() (A' .. F).
//
// Schedule: static (tile += G/8) or, with d.dynamic, dynamic: after its first tile a workgroup
// draws tile numbers G/8 + n from its XCD's counter.  Thread 0 asks one tile ahead, so the
// atomic's round trip is waited for together with the prefetched window (same vmcnt).  It is an
// atomicInc, not atomicAdd: LLVM would aggregate an add over the wave and wait for its result at once.
//
// Everything derived from the thread index is recomputed per tile from an opaque copy (the empty
// asm): hoisted out of the tile loop those values cost more registers than the 64 that eight
// waves per SIMD allow, and a spill reload (vmcnt) would wait for the prefetch it sits behind.
template <int NV, class Kern>
__device__ __forceinline__ void persistent_tiles(const SeaDev& d, uint32_t* lds, const Layout L)
{
    const int xcd = blockIdx.x & 7, gx = gridDim.x >> 3;
    const int npairs_x = (d.pairs - xcd + 7) >> 3;          // pairs with pair % 8 == xcd
    const int ntiles = npairs_x * d.wg_per_pair;
    int tile = blockIdx.x >> 3;
    if (tile >= ntiles) return;

    uint32_t wv[NV], an_next = 0;
    int pair = 0, trow = 0, bcol0 = 0;
    auto fetch = [&](int t) {
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        const int row0 = div_small(tid, d.magic_pitch), dw = tid - row0 * d.pitch_dw;
        const int lp = (int)(((unsigned long long)(unsigned)t * d.magic_wpp) >> 40);
        const int wg = t - lp * d.wg_per_pair;
        trow = (int)(((unsigned long long)(unsigned)wg * d.magic_wpr) >> 40);
        bcol0 = (wg - trow * d.wg_per_row) * d.tc;
        pair = lp * 8 + xcd;
        // Window rows through a buffer resource over the pair's `cur` plane (H * pitch bytes): rows above or below
        // the frame give offsets outside it and the hardware range check returns 0 -- what the search wants there
        // (bbme.py:157-162 skips such candidates; they never form keys) -- so no per-row guard or branch is left.
        // Columns outside the plane and threads beyond the staging sweep get an offset that is out of range by itself.
        const uint8_t* cur = d.cur + (long long)pair * d.plane_stride;
        const unsigned long long cur_bits = (unsigned long long)cur;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(cur_bits >> 32)) << 32) |
                    (unsigned)__builtin_amdgcn_readfirstlane((int)cur_bits)),
            (short)0, __builtin_amdgcn_readfirstlane(d.H * d.pitch), 0x00020000);
        const int gx0 = bcol0 * 16 - d.sw + 4 * dw, gy0 = trow * d.tr * 16 - d.sw + row0;
        const bool colok = row0 < d.rstep && gx0 >= 0 && gx0 < d.pitch;
        const int off0 = colok ? gy0 * d.pitch + gx0 : (int)0x80000000;
        const int sstep = d.rstep * d.pitch;
#pragma unroll
        for (int u = 0; u < NV; ++u)
            wv[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, colok ? off0 + u * sstep : off0, 0, 0);
        an_next = 0;
        const WaveBlock wb = wave_block(d, trow, bcol0, wave);
        if (wb.ok) {
            const uint8_t* aptr = d.prev + (long long)pair * d.plane_stride + (long long)(wb.brow * 16) * d.pitch + wb.bcol * 16;
            an_next = *(const uint32_t*)(aptr + (long long)(lane >> 2) * d.pitch + (lane & 3) * 4);
        }
    };
    fetch(tile);
    uint32_t* ctr = d.dynamic ? d.status + GME_STATUS_TILECTR + 16 * xcd : nullptr;
    uint32_t drawn = 0;
    if (ctr && threadIdx.x == 0) drawn = atomicInc(ctr, 0xFFFFFFFFu);
    if (threadIdx.x == 0) { lds[L.count] = 0; lds[L.count + 3] = 0; lds[L.count + 4] = 0; lds[L.count + 5] = 0; lds[L.count + 7] = 0; }
    uint32_t stat_scored = 0, stat_listed = 0;             // thread 0's: patches scored exactly / left by the first upper bounds
    for (;;) {
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        {   // registers -> LDS
            const int row0 = div_small(tid, d.magic_pitch);
            if (row0 < d.rstep) {
                uint32_t* dst = lds + L.win + tid;
                const int dstep = d.rstep * d.pitch_dw;
#pragma unroll
                for (int u = 0; u < NV; ++u)
                    if (row0 + u * d.rstep < d.win_rows) dst[u * dstep] = wv[u];
            }
        }
        const int pair_c = pair, trow_c = trow, bcol0_c = bcol0, tile_c = tile;
        const uint32_t mine = an_next;
        const typename Kern::Pre pre = Kern::prep(d, lds, L, wave, lane, wave_block(d, trow_c, bcol0_c, wave).ok, mine);
        if (tid == 0) {
            // the finished tile's counts (final behind its phase D barrier) -> statistics, then cleared for this tile
            const uint32_t listed = lds[L.count], c2_off = lds[L.count + 4], c2_scored = lds[L.count + 5];
            stat_scored += listed + c2_scored;
            stat_listed += listed + c2_off;
            lds[L.count] = 0; lds[L.count + 4] = 0; lds[L.count + 5] = 0; lds[L.count + 7] = 0;
            if (ctr) lds[L.count + 1] = (uint32_t)gx + drawn;
        }
        __syncthreads();
        tile = ctr ? (int)lds[L.count + 1] : tile + gx;
        const bool more = tile < ntiles;                   // workgroup-uniform
        if (more) {
            fetch(tile);
            if (ctr && tid == 0) drawn = atomicInc(ctr, 0xFFFFFFFFu);
        }
#ifdef SEA_NO_REDO
        Kern::phases(d, lds, L, pair_c, trow_c, bcol0_c, mine, pre, tid, tile_c);
#else
        // A hostile tile is handed to the redo kernel.  When the tile this workgroup processed just before was
        // hostile too (a scene cut, a noisy shot: not an isolated occlusion), thread 0's wave also draws the next
        // tile numbers of this XCD -- consecutive tiles of the same pairs -- and lists them unseen: 3 of them, then
        // 7, then REDO_BURST while the streak lasts.  Where every tile is hostile only one in REDO_BURST + 1 pays
        // for the bound phases before brute force does the work anyway; friendly content never takes the branch.
        if (!Kern::phases(d, lds, L, pair_c, trow_c, bcol0_c, mine, pre, tid, tile_c)) {
            if (tid == 0) lds[L.count + 3] = 0;            // streak of hostile tiles (thread 0's word)
        } else if (ctr && wave == 0) {
            const int streak = (int)lds[L.count + 3];      // same address for the whole wave; only thread 0 writes it
            const int burst = streak == 0 ? 0 : streak == 1 ? 3 : streak == 2 ? 7 : REDO_BURST;
            if (burst) {
                // lane 0 lists the tile it had drawn already, lanes 1 .. burst draw one each (one wave-aggregated
                // atomic); the last of those becomes thread 0's new `drawn`, the others are listed
                uint32_t got = 0;
                if (lane >= 1 && lane <= burst) got = atomicAdd(ctr, 1u);
                const int t = gx + (int)(lane == 0 ? drawn : got);
                if (lane < burst && t < ntiles) push_redo(d, t, xcd);
                drawn = (uint32_t)__shfl((int)got, burst, 64);
            }
            if (lane == 0) lds[L.count + 3] = (uint32_t)(streak + 1);
        }
#endif
        if (!more) break;
        // No barrier here (round 4): every wave has passed phase D's barrier, and phase E's chunks end in one each, so
        // nobody still reads the window, the anchors, the box sums or the list when the next tile's staging overwrites them;
        // what is touched between here and the next tile's first barrier is per wave (best[], prev[]) or thread 0's
        // (the count words: a wave that has not read the list length yet can only be in a tile whose list is empty).
#ifdef SEA_TILE_END_BARRIER
        __syncthreads();
#endif
    }
    // thread 0 has passed the barrier behind phase D: the last tile's counts are final
    if (threadIdx.x == 0) {
        atomicAdd(d.status + GME_STATUS_STATS + 16 * xcd, stat_scored + lds[L.count] + lds[L.count + 5]);
        atomicAdd(d.status + GME_STATUS_STATS + 16 * xcd + 2, stat_listed + lds[L.count] + lds[L.count + 4]);
    }
}

// Host side of the persistent form: resident workgroups per XCD (what LDS and the 32 wave slots of
// a CU allow, on every CU) and whether the launch qualifies.  GME_SEA_PERSIST = 0 forces the
// one-tile-per-workgroup kernel, 1 / 2 force the persistent one (static / dynamic schedule);
// unset: persistent with the dynamic schedule once every resident workgroup gets several tiles.
struct PersistPlan { bool use; bool dynamic; long long g; int nv; };
inline PersistPlan plan_persistent(const SeaDev& d, size_t lds_bytes, int pairs, int cu_count)
{
    PersistPlan p;
    const long long tiles_x = (((long long)pairs + 7) / 8) * d.wg_per_pair;          // per XCD
    p.nv = (d.win_rows + d.rstep - 1) / d.rstep;
    const char* pmode = getenv("GME_SEA_PERSIST");
    const int pm = pmode ? atoi(pmode) : -1;
    int per_cu = (int)((160 * 1024) / (lds_bytes + 1024));
    if (per_cu * d.nb > 32) per_cu = 32 / d.nb;
    if (per_cu < 1) per_cu = 1;
    p.g = (long long)per_cu * cu_count / 8;
    const bool can = tiles_x < (1ll << 21) && d.wg_per_pair < (1 << 18) && p.nv <= 16 && p.g >= 1;
    p.use = can && (pm > 0 || (pm < 0 && tiles_x >= 4 * p.g));
    p.dynamic = pm != 1;
    if (p.g > tiles_x) p.g = tiles_x;
    return p;
}

}  // namespace sea
