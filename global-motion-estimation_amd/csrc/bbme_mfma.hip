// Exhaustive block matching, bs = 16, MSE, on the matrix cores (bbme.py:105-179 with pnorm 1; BASELINE configs[3]).
//
// SSD(dy, dx) = sum(a^2) + sum(w^2) - 2 sum_{r,c} a[r][c] w[r + dy][c + dx]: the cross term is a correlation of the 16x16
// anchor with the search window, and exhaustive search wants it for EVERY displacement -- work that is GEMM-shaped as it
// stands (nothing is reshaped to get there) and that the vector unit pays one v_dot4 per four byte pairs for
// (k_exh_dot16, and phase E of the elimination kernel: 92 vector instructions per surviving candidate).
// Here one v_mfma_i32_16x16x64_i8 scores 16 x 16 displacements against 4 anchor rows x 16 window columns:
//
//   C[dy][dx] = sum_r sum_X  W[r + dy][X] * a[r][X - dx]         X = window column, c = X - dx the anchor column
//
//   A operand (M = dy, K = (4 rows, 16 columns)): lane (m = l & 15, g = l >> 4) holds W[d0 + m + r0 + g][X0 .. X0 + 15] --
//       sixteen bytes of ONE window row at a 16-byte-aligned column: one conflict-free ds_read_b128 from the tile's window
//       in the LDS (row pitch 16 * odd bytes).  It depends on d0 + r0 and X0 only: (NC / 4) x (NT + 1) operands per block.
//   B operand (K, N = dx): lane (n = l & 15, g) holds a[r0 + g][X0 + j - x0 - n], j = 0 .. 15: anchor row r0 + g shifted
//       right by n bytes (X0 = x0) or left by 16 - n (X0 = x0 + 16), zeros elsewhere -- 8 operands per block, cut out of a
//       zero-padded image of the anchor in the LDS once per block and kept in 32 VGPRs.
//   The k index of both operands is (g, j) in the same lane positions, so the products pair up whatever order the
//   instruction walks k in; C comes back as col = l & 15 (dx), row = 4 (l >> 4) + register (dy).
//
// Bytes are made int8 by flipping the top bit (b - 128): differences are unchanged, so
//   SSD = sum(a'^2) + sum(w'^2) - 2 C'   with a' = a - 128, w' = w - 128,
// sum(w'^2) from the per-frame table of 16x16 box sums (k_sqbox16<true>, aux kind 2), sum(a'^2) once per block.  All terms
// are exact integers < 2^24, equal to the reference's float32 sums (bbme.py:94).  The winner is the first strict minimum in
// the reference's scan order (column offset outer, bbme.py:146-149,171): per lane the candidates are visited in ascending
// scan order under a (cost << 7 | local index) key, the wave then takes the smallest cost and, among its holders, the
// smallest scan index.  Every candidate is scored: run time does not depend on the content (no elimination, no redo pass).
//
// One wave per macroblock, a workgroup = TC horizontally adjacent blocks sharing one window (small workgroups: the chain
// stage -> barrier -> correlate -> costs is serial inside one, so a CU overlaps many of them); 3-D grid without a division
// (x = tile column * 8 + XCD slot, y = block row, z = group of 8 pairs: the 8 pairs of a group run on the 8 XCDs, a pair's
// frames enter one L2).  Per block at sw 16: 72 MFMAs (1152 cycles of the matrix pipe), 48 ds_read_b128, 36 table reads.
#include <stdio.h>
#include <stdlib.h>

#include "gme_internal.h"

// search windows the kernel takes without being asked (GME_EXH_MFMA unset): up to this one.  At sw 32 it is content-independent
// (31 k pairs/s at 1080p on noise against 6.3 k) but 8 % behind the elimination kernel on the synthetic pan of configs[3]
// (31.5 k against 34.4 k): opt-in there (GME_EXH_MFMA=1).
#ifndef GME_EXH_MFMA_AUTO_SW
#define GME_EXH_MFMA_AUTO_SW 16
#endif
// blocks per workgroup (sw <= 16 / wider), see the sweep in DESIGN.md
#ifndef MFMA_TC_SMALL
#define MFMA_TC_SMALL 4
#endif
#ifndef MFMA_TC_LARGE
#define MFMA_TC_LARGE 4
#endif
// table reads of a pass's first tile column before the correlation (costs 12-20 registers = a wave per SIMD: slower, measured)
#ifndef MFMA_EARLY_TABLE
#define MFMA_EARLY_TABLE 0
#endif
// the next anchor row group's B-operand dwords read a row group ahead: 10 registers, 88-95 VGPRs instead of 78-81 = 5
// instead of 6 waves per SIMD at sw 16 (same box, 1x4 tiles: 397.8 k against 428.9 k pairs/s with it off)
#ifndef MFMA_BOP_PREFETCH
#define MFMA_BOP_PREFETCH 0
#endif
// Waves per SIMD decide this kernel (the unified file holds VGPRs + AGPRs; the accumulators live in AGPRs).  At sw <= 16:
// window operands single-buffered and the accumulator tiles pinned by an empty asm after every step -> 45 + 48 registers =
// 5 waves (double-buffered, unpinned: 77 + 48 = 4 waves; same box 415.4 k -> 429.1 k pairs/s).  At sw 32 neither form gets
// under the 128 registers of 4 waves and double buffering is worth 5 % (32.7 k against 31.1 k): -1 = that choice by NT.
#ifndef MFMA_PIN_ACC
#define MFMA_PIN_ACC -1
#endif
#ifndef MFMA_WQ_DOUBLE
#define MFMA_WQ_DOUBLE -1
#endif
#ifndef MFMA_STAGE_PREFETCH
#define MFMA_STAGE_PREFETCH 1
#endif
// tile columns per pass (XS) at sw <= 16 / wider
#ifndef MFMA_XS_SMALL
#define MFMA_XS_SMALL 3
#endif
#ifndef MFMA_XS_LARGE
#define MFMA_XS_LARGE 2
#endif
// block rows a workgroup walks (1: 386 k, 6: 404-421 k, 30: 409-421 k pairs/s at 720x480, same box)
#ifndef MFMA_ROWS_PER_WG
#define MFMA_ROWS_PER_WG 6
#endif

namespace {

struct MfmaDev {
    const uint8_t* prev;
    const uint8_t* cur;
    long long plane_stride;
    int pairs, H, W, pitch, sw;
    int nbr, nbc;
    int rpw;                          // block rows per workgroup
    int32_t* mf;
    const uint32_t* sq;               // 16x16 box sums of (byte - 128)^2 of `cur`, [pairs][H][pitch]
    long long sq_stride;
};

typedef int v4i __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int ANCH_ROW_DW = 12;                   // 16 zero bytes | 16 anchor bytes | 16 zero bytes
constexpr int ANCH_DW = 16 * ANCH_ROW_DW + 8;     // + slack for the fifth dword of the last row's reads
constexpr uint32_t FLIP = 0x80808080u;

// LDS window of a tile of TC blocks: NC + 15 rows, row pitch 16 * odd bytes -- the 16 rows of one ds_read_b128 then fall
// on 16 distinct bank quads
template <int NT, int TC> struct MfmaGeo {
    static constexpr int NC = 16 * NT;
    static constexpr int ROWS = NC + 15;
    static constexpr int RAW = 16 * (TC - 1) + NC + 16;
    static constexpr int PITCH = ((RAW / 16) & 1) ? RAW : RAW + 16;
    static constexpr int LDS_BYTES = ROWS * PITCH + TC * ANCH_DW * 4;
};

// What a pass needs of its block (all but `key` wave- or lane-constant)
struct MfmaBlock {
    const uint8_t* wbase;             // lane's window address: row m + g, column (block's window column 0)
    const uint32_t* anch;             // the wave's anchor image
    int pitch;
    int n, g;
    int lo_r, hi_r, lo_c, hi_c;       // candidate indices inside the frame
    int tlane;                        // byte offset of the table position of candidate (row index 4g, column index n)
    int tuni, tvar;                   // the same split into a wave-uniform part (block) and the lane's part (interior blocks)
    bool interior;                    // every candidate of the block lies inside the frame (wave-uniform)
};

// the five dwords that hold the lane's 16 bytes of B operand (kc, rg), and the operand cut out of them
__device__ __forceinline__ void bop_read(const MfmaBlock& k, int kc, int rg, uint32_t (&l)[5])
{
    const int off = (4 * rg + k.g) * (4 * ANCH_ROW_DW) + 16 + 16 * kc - k.n;      // >= 1
    const uint32_t* p = k.anch + (off >> 2);
#pragma unroll
    for (int j = 0; j < 5; ++j) l[j] = p[j];
}
__device__ __forceinline__ v4i bop_cut(const MfmaBlock& k, int kc, const uint32_t (&l)[5])
{
    const uint32_t sh = (uint32_t)(16 * kc - k.n) & 3u;                            // = off & 3
    v4i t;
    t.x = (int)__builtin_amdgcn_alignbyte(l[1], l[0], sh);
    t.y = (int)__builtin_amdgcn_alignbyte(l[2], l[1], sh);
    t.z = (int)__builtin_amdgcn_alignbyte(l[3], l[2], sh);
    t.w = (int)__builtin_amdgcn_alignbyte(l[4], l[3], sh);
    return t;
}

// One pass: the tile columns TX0 .. TX0 + TXN - 1 (16 column offsets each) of all NT tile rows -- NT x TXN accumulators.
// Steps run anchor row group (rg) outer, tile row (ty) inner, so that only the two B operands of one row group are live;
// the TXN + 1 window operands of a step are read while the MFMAs of the step before run (every LDS offset is an
// immediate: PITCH is a constant of the instance).  Then the costs in ascending scan order under signed keys
// ((sum(w'^2) - 2 C') << 7 | local index; the block's sum(a'^2) is added once, at the end): interior blocks -- all
// candidates valid, the common case -- pay 3 vector instructions per candidate and read the table through a scalar
// offset, blocks at a frame edge test every candidate and read through a range-checked lane offset.
template <int NT, int PITCH, int TX0, int TXN>
__device__ __forceinline__ void mfma_pass(const MfmaBlock& k, const __amdgpu_buffer_rsrc_t rs, int& key)
{
    // table reads of one tile (tx, ty): interior blocks through a scalar offset, edge blocks through a range-checked lane offset
    uint32_t tb[2][4];
    int inner = 0;
    auto table_reads = [&](int t, uint32_t (&out)[4]) {           // t = tx * NT + ty: ascending scan order
        const int tx = t / NT, ty = t - tx * NT;
        if (inner) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                out[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, k.tvar + 64 * (TX0 + tx), k.tuni + (16 * ty + i) * k.pitch * 4, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                out[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, k.tlane + ((16 * ty + i) * k.pitch + 16 * (TX0 + tx)) * 4, 0, 0);
        }
    };
    if (MFMA_EARLY_TABLE) table_reads(0, tb[0]);
    constexpr int STEPS = 4 * NT;
    constexpr bool WQ_DOUBLE = MFMA_WQ_DOUBLE < 0 ? NT > 3 : MFMA_WQ_DOUBLE != 0;
    constexpr bool PIN_ACC = MFMA_PIN_ACC < 0 ? NT <= 3 : MFMA_PIN_ACC != 0;
    v4i acc[NT][TXN];
#pragma unroll
    for (int ty = 0; ty < NT; ++ty)
#pragma unroll
        for (int tx = 0; tx < TXN; ++tx) acc[ty][tx] = v4i{0, 0, 0, 0};
    u32x4 wq[2][TXN + 1];
    const uint8_t* wcol = k.wbase + 16 * TX0;
    if (WQ_DOUBLE) {
#pragma unroll
        for (int xi = 0; xi <= TXN; ++xi) wq[0][xi] = *(const u32x4*)(wcol + 16 * xi);      // step 0: rg 0, ty 0
    }
    uint32_t raw[2][5];
    bop_read(k, 0, 0, raw[0]);
    bop_read(k, 1, 0, raw[1]);
    v4i bop[2] = { bop_cut(k, 0, raw[0]), bop_cut(k, 1, raw[1]) };
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        const int rg = st / NT, ty = st - rg * NT;
        if (!WQ_DOUBLE) {
            const uint8_t* wrow = wcol + 4 * (4 * ty + rg) * PITCH;
#pragma unroll
            for (int xi = 0; xi <= TXN; ++xi) wq[st & 1][xi] = *(const u32x4*)(wrow + 16 * xi);
        }
        if (WQ_DOUBLE && st + 1 < STEPS) {
            const int rg1 = (st + 1) / NT, ty1 = (st + 1) - rg1 * NT;
            const uint8_t* wrow = wcol + 4 * (4 * ty1 + rg1) * PITCH;
#pragma unroll
            for (int xi = 0; xi <= TXN; ++xi) wq[(st + 1) & 1][xi] = *(const u32x4*)(wrow + 16 * xi);
            if (MFMA_BOP_PREFETCH && ty == 0 && rg + 1 < 4) { bop_read(k, 0, rg + 1, raw[0]); bop_read(k, 1, rg + 1, raw[1]); }
        }
        if (WQ_DOUBLE) __builtin_amdgcn_sched_barrier(0);             // the next step's reads are in flight before this one's MFMAs
#pragma unroll
        for (int xi = 0; xi <= TXN; ++xi) {
            const u32x4 t = wq[st & 1][xi];
            const v4i wop = { (int)t.x, (int)t.y, (int)t.z, (int)t.w };
            if (xi < TXN) acc[ty][xi] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wop, bop[0], acc[ty][xi], 0, 0, 0);
            if (xi >= 1) acc[ty][xi - 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wop, bop[1], acc[ty][xi - 1], 0, 0, 0);
        }
        if (PIN_ACC) {
#pragma unroll
            for (int xi = 0; xi < TXN; ++xi) asm volatile("" : "+a"(acc[ty][xi]));
        }
        if (ty == NT - 1 && rg + 1 < 4) {
            if (!MFMA_BOP_PREFETCH) { bop_read(k, 0, rg + 1, raw[0]); bop_read(k, 1, rg + 1, raw[1]); }
            bop[0] = bop_cut(k, 0, raw[0]); bop[1] = bop_cut(k, 1, raw[1]);
        }
    }
    // ---- costs; first minimum in scan order (column index outer, row index inner).  The table reads of the next tile
    //      are in flight while the keys of one tile are formed (four registers each: the accumulators sit in AGPRs and
    //      come through v_accvgpr_read one tile at a time; a whole tile column in flight cost a wave per SIMD).
    // (the flag is made opaque here: known before the correlation, the compiler duplicated the loop's last MFMAs into
    //  both branches with 12 more accumulator registers -- a wave per SIMD)
    inner = k.interior ? 1 : 0;
    asm volatile("" : "+s"(inner));
    const bool interior = inner != 0;
    if (!MFMA_EARLY_TABLE) table_reads(0, tb[0]);
#pragma unroll
    for (int t = 0; t < TXN * NT; ++t) {
        const int tx = t / NT, ty = t - tx * NT;
        if (t + 1 < TXN * NT) table_reads(t + 1, tb[(t + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const int acc4[4] = { acc[ty][tx].x, acc[ty][tx].y, acc[ty][tx].z, acc[ty][tx].w };
        if (interior) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cost = __mul24(acc4[i], -2) + (int)tb[t & 1][i];               // |C'| < 2^23: one v_mad_i32_i24
                int kk;                                                                  // (cost << 7) | local: gfx9 has no VOP3 literals,
                asm("v_lshl_or_b32 %0, %1, 7, %2" : "=v"(kk) : "v"(cost), "s"(((TX0 + tx) * NT + ty) * 4 + i));   // the index rides in an SGPR
                key = min(key, kk);
            }
        } else {
            const int ci = 16 * (TX0 + tx) + k.n;
            const bool cok = ci >= k.lo_c && ci <= k.hi_c;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ri = 16 * ty + 4 * k.g + i;
                const int cost = (int)tb[t & 1][i] - 2 * acc4[i];
                const int kk = (int)(((uint32_t)cost << 7) | (uint32_t)(((TX0 + tx) * NT + ty) * 4 + i));
                if (cok && ri >= k.lo_r && ri <= k.hi_r) key = min(key, kk);
            }
        }
    }
}

// XS = tile columns per pass: NT x XS x 4 accumulator registers (sw 32: 5 x 3 and 5 x 2 instead of 5 x 5 = 100).
// A workgroup walks d.rpw consecutive block rows of its tile column: the next row's window and anchors are fetched into
// registers while the current one is searched (global latency behind the MFMAs, 1 / rpw of the workgroup launches).
// Registers decide this kernel's speed (latencies are covered by waves, not by one wave's schedule).  An occupancy
// attribute does not help -- asked for 5 or 6 waves the scheduler first builds its usual pressure and then spills 33-67
// registers; what keeps the count down is in the code: row groups outer (two B operands live), no operand prefetch, the
// opaque lane indices of the row loop, one tile's table reads at a time, pinned accumulators (see MFMA_PIN_ACC).
// tests/test_host.py holds the sw 16 instance to 5 waves.
template <int NT, int XS, int TC>
__global__ void __launch_bounds__(64 * TC) k_exh_mfma16(MfmaDev d)
{
    typedef MfmaGeo<NT, TC> G;
    constexpr int NC = G::NC, PITCH = G::PITCH, PDW = PITCH / 4, P8 = PITCH / 8;
    constexpr int RSTEP = 64 * TC / P8, NST = (G::ROWS + RSTEP - 1) / RSTEP;
    __shared__ __attribute__((aligned(16))) uint32_t lds[G::LDS_BYTES / 4];
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = (int)blockIdx.z * 8 + ((int)blockIdx.x & 7);
    if (pair >= d.pairs) return;
    const int brow0 = (int)blockIdx.y * d.rpw, bc0 = ((int)blockIdx.x >> 3) * TC;
    const int nrows = min(d.rpw, d.nbr - brow0);
    uint32_t* win = lds;
    uint32_t* anch = lds + G::ROWS * PDW + wave * ANCH_DW;
    const int bcol = bc0 + wave;
    const bool mine = bcol < d.nbc;                                   // ragged last tile: the wave only helps staging
    const int c0 = bcol * 16;
    const uint8_t* cur = d.cur + (long long)pair * d.plane_stride;
    const uint8_t* prev = d.prev + (long long)pair * d.plane_stride;

    // ---- staging: the tile's window, rows r0 - sw .., columns bc0*16 - sw .. (8 bytes per thread and row; out of the
    //      frame: anything, those candidates are not scored), and the wave's anchor dword (row lane / 4, dword lane % 4)
    const int srow0 = tid / P8, c8 = tid - srow0 * P8;
    const int gx = bc0 * 16 - d.sw + 8 * c8;                          // % 8 == 0 (sw % 8 == 0)
    const bool colok = srow0 < RSTEP && gx >= 0 && gx < d.pitch;      // pitch % 64 == 0: the 8 bytes are inside the row or not at all
    uint2 sv[NST];
    uint32_t av = 0;
    auto stage_load = [&](int brow) {
        const int gy0 = brow * 16 - d.sw + srow0;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int gy = gy0 + i * RSTEP;
            sv[i] = make_uint2(0u, 0u);
            if (colok && gy >= 0 && gy < d.H && srow0 + i * RSTEP < G::ROWS) sv[i] = *(const uint2*)(cur + (long long)gy * d.pitch + gx);
        }
        if (mine) av = *(const uint32_t*)(prev + (long long)(brow * 16 + (lane >> 2)) * d.pitch + c0 + 4 * (lane & 3));
    };
    MfmaBlock k;
    int ln = lane & 15, lg = lane >> 4;
    k.anch = anch; k.pitch = d.pitch;
    k.lo_c = max(0, d.sw - c0); k.hi_c = min(NC - 1, d.W - 16 - c0 + d.sw);
    const uint32_t* tab = d.sq + (long long)pair * d.sq_stride;
    const uint64_t tbits = (uint64_t)tab;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(tbits >> 32)) << 32) |
                (uint32_t)__builtin_amdgcn_readfirstlane((int)tbits)),
        (short)0, __builtin_amdgcn_readfirstlane(d.H * d.pitch * 4), 0x00020000);

    stage_load(brow0);
    for (int it = 0; it < nrows; ++it) {
        const int brow = brow0 + it, r0 = brow * 16;
        if (srow0 < RSTEP) {
#pragma unroll
            for (int i = 0; i < NST; ++i)
                if (srow0 + i * RSTEP < G::ROWS) {
                    uint2 v = sv[i];
                    v.x ^= FLIP; v.y ^= FLIP;
                    *(uint2*)(win + (srow0 + i * RSTEP) * PDW + 2 * c8) = v;
                }
        }
        uint32_t a2 = 0;
        if (mine) {
            const uint32_t a8 = av ^ FLIP;
            uint32_t* row = anch + (lane >> 2) * ANCH_ROW_DW + (lane & 3);
            row[0] = 0; row[4] = a8; row[8] = 0;
            if (lane < 8) anch[16 * ANCH_ROW_DW + lane] = 0;
            a2 = wave_sum_u32(__builtin_amdgcn_udot4(a8 ^ FLIP, a8 ^ FLIP, 0u, false)) - 256u * wave_sum_u32(__builtin_amdgcn_udot4(a8 ^ FLIP, 0x01010101u, 0u, false)) + 256u * 16384u;
        }
        __syncthreads();
        if (MFMA_STAGE_PREFETCH && it + 1 < nrows) stage_load(brow + 1);     // in flight while this row is searched
        if (mine) {
            // opaque per row: everything derived from the lane's (n, g) -- LDS addresses, shift amounts, table offsets, some
            // 40 registers -- would otherwise be hoisted out of the row loop and held across it (136 VGPRs instead of ~80)
            asm volatile("" : "+v"(ln), "+v"(lg));
            k.n = ln; k.g = lg;
            k.wbase = (const uint8_t*)win + (k.n + k.g) * PITCH + wave * 16;
            k.tvar = (4 * k.g * d.pitch + k.n) * 4;
            k.lo_r = max(0, d.sw - r0); k.hi_r = min(NC - 1, d.H - 16 - r0 + d.sw);
            k.interior = __builtin_amdgcn_readfirstlane(k.lo_r == 0 && k.lo_c == 0 && k.hi_r == NC - 1 && k.hi_c == NC - 1);
            // negative = out of range as unsigned: the buffer read returns 0, the candidate is not scored
            k.tuni = ((r0 - d.sw) * d.pitch + (c0 - d.sw)) * 4;
            k.tlane = k.tuni + k.tvar;
            int key = 0x7FFFFFFF;
            mfma_pass<NT, PITCH, 0, (XS < NT ? XS : NT)>(k, rs, key);
            if constexpr (XS < NT) { __builtin_amdgcn_sched_barrier(0); mfma_pass<NT, PITCH, XS, (2 * XS < NT ? XS : NT - XS)>(k, rs, key); }
            if constexpr (2 * XS < NT) { __builtin_amdgcn_sched_barrier(0); mfma_pass<NT, PITCH, 2 * XS, NT - 2 * XS>(k, rs, key); }
            static_assert(3 * XS >= NT, "at most three passes");

            const int n = k.n, g = k.g;
            const uint32_t bcost = key == 0x7FFFFFFF ? 0xFFFFFFFFu : (uint32_t)((key >> 7) + (int)a2);      // the SSD: 0 .. 2^24
            const uint32_t cmin = wave_min_u32(bcost);
            uint32_t myidx = 0xFFFFFFFFu;
            if (bcost == cmin && key != 0x7FFFFFFF) {
                const int local = key & 127;
                const int i = local & 3, t = local >> 2;
                const int tx = t / NT, ty = t - tx * NT;
                myidx = (uint32_t)((16 * tx + n) * NC + 16 * ty + 4 * g + i);
            }
            const uint32_t best = wave_min_u32(myidx);
            if (lane == 0) {
                const int idx = (int)(best & 0x1FFF);
                const int ci = idx / NC, ri = idx - ci * NC;
                int32_t* o = d.mf + (((long long)pair * d.nbr + brow) * d.nbc + bcol) * 2;
                o[0] = ci - d.sw;
                o[1] = ri - d.sw;
            }
        }
        if (it + 1 < nrows) {
            __syncthreads();                                          // every wave has read the window before it is overwritten
            if (!MFMA_STAGE_PREFETCH) stage_load(brow + 1);
        }
    }
}

template <int NT, int XS, int TC>
void mfma_launch(gme_ctx* ctx, const MfmaDev& d)
{
    const dim3 grid((unsigned)((d.nbc + TC - 1) / TC) * 8, (unsigned)((d.nbr + d.rpw - 1) / d.rpw), (unsigned)((d.pairs + 7) / 8)), block(64 * TC);
    plan_note(ctx, 0, "k_exh_mfma16<%d> 1x%d blocks per workgroup, %d rows each, grid %ux%ux%u", NT, TC, d.rpw, grid.x, grid.y, grid.z);
    hipLaunchKernelGGL((k_exh_mfma16<NT, XS, TC>), grid, block, 0, ctx->stream, d);
}

}  // namespace

// Search windows the matrix-core kernel takes: NC = 2 sw + 16 a multiple of 16, at most 5 tiles per axis.
// GME_EXH_MFMA=0 keeps exhaustive MSE on the vector unit (the elimination kernels of bbme_sea_mse.hip), =1 takes every
// window this kernel can, unset: windows up to GME_EXH_MFMA_AUTO_SW.
bool bbme_mfma_wanted(int sw)
{
    if (sw < 0 || sw % 8 != 0 || sw > 32) return false;
    if (getenv("GME_FORCE_GENERIC") || getenv("GME_EXH_BRUTE")) return false;
    const char* e = getenv("GME_EXH_MFMA");
    return e ? atoi(e) != 0 : sw <= GME_EXH_MFMA_AUTO_SW;
}

int launch_bbme_mfma(gme_ctx* ctx, const BbmeJob& job, bool* handled)
{
    *handled = false;
    if (job.procedure != GME_SEARCH_EXHAUSTIVE || job.bs != 16 || job.pnorm != GME_NORM_MSE) return GME_OK;
    if (job.sqbox_cur == nullptr || !bbme_mfma_wanted(job.sw)) return GME_OK;
    if (bbme_aux_kind(job.bs, job.sw, job.procedure, job.pnorm) != 2) return GME_OK;
    const int NC = 2 * job.sw + 16, NT = NC / 16;
    const int nbr = job.H / 16, nbc = job.W / 16;
    if (nbr == 0 || nbc == 0 || nbr > 65535 || (job.pairs + 7) / 8 > 65535) return GME_OK;
    if ((long long)(job.H + 64) * job.pitch * 4 >= (1ll << 31)) return GME_OK;          // 32-bit table offsets

    MfmaDev d;
    d.prev = job.prev; d.cur = job.cur; d.plane_stride = job.plane_stride;
    d.pairs = job.pairs; d.H = job.H; d.W = job.W; d.pitch = job.pitch; d.sw = job.sw;
    d.nbr = nbr; d.nbc = nbc; d.mf = job.mf;
    d.sq = job.sqbox_cur; d.sq_stride = job.sqbox_stride;
    d.rpw = getenv("GME_MFMA_ROWS") ? atoi(getenv("GME_MFMA_ROWS")) : MFMA_ROWS_PER_WG;
    if (d.rpw < 1) d.rpw = 1;
    if (d.rpw > nbr) d.rpw = nbr;
    int tc = NT <= 3 ? MFMA_TC_SMALL : MFMA_TC_LARGE;
    if (const char* e = getenv("GME_MFMA_TILE")) { int a = 0, c = 0; if (sscanf(e, "%dx%d", &a, &c) == 2 && a == 1 && c >= 1 && c <= 4) tc = c; }
#define MFMA_CASE(NTV, XSV) \
    case NTV: if (tc == 1) mfma_launch<NTV, XSV, 1>(ctx, d); else if (tc == 2) mfma_launch<NTV, XSV, 2>(ctx, d); \
              else if (tc == 3) mfma_launch<NTV, XSV, 3>(ctx, d); else mfma_launch<NTV, XSV, 4>(ctx, d); break
    switch (NT) {
    MFMA_CASE(1, 1);
    MFMA_CASE(2, 2);
    MFMA_CASE(3, MFMA_XS_SMALL);
    MFMA_CASE(4, 2);
    default: MFMA_CASE(5, MFMA_XS_LARGE);
    }
#undef MFMA_CASE
    GME_HIP_TRY(hipGetLastError());
    *handled = true;
    return GME_OK;
}
