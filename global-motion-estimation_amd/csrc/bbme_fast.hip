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
#include <stdlib.h>

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
    const uint32_t* sqbox;        // MSE: 16x16 box sums of squares of `cur`, [pairs][H][pitch]
    long long sqbox_stride;       // elements between consecutive planes
};

typedef uint64_t u64_a4 __attribute__((aligned(4)));
typedef uint32_t u32x4_v __attribute__((ext_vector_type(4)));
// The anchor block is wave-uniform and the frames are never written by a search kernel: read through the constant
// address space its 64 dwords become scalar loads (SGPRs) that no store or atomic of the kernel can alias -- also inside
// the redo kernel's work loop, where plain global loads ended up in 86-128 VGPRs.
typedef __attribute__((address_space(4))) const uint32_t const_u32;
__device__ __forceinline__ const_u32* as_constant(const void* p) { return (const_u32*)(uintptr_t)p; }

// ---- stage the search window (coalesced dword loads; out-of-frame -> 0) ----
// thread -> one dword column and every `rstep`-th row: no div/mod inside the loop
__device__ __forceinline__ void stage_window(const FastDev& d, uint32_t* win, const uint8_t* cur, int bcol0, int r0, int tid)
{
    const int gx0 = bcol0 * 16 - d.sw;             // multiple of 4 (sw % 4 == 0)
    const int gy0 = r0 - d.sw;
    const int rstep = blockDim.x / d.pitch_dw;
    const int row0 = tid / d.pitch_dw, dw = tid - row0 * d.pitch_dw;
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

// The search of one macroblock by one wave; the window of the workgroup's blocks is staged in `win`.
// ALDS: the anchors of the workgroup's blocks are staged in LDS (`alds`, 64 dwords per wave) and each anchor
// row is read just before its first use (one broadcast ds_read_b128, R rows live) instead of sitting in 64
// SGPRs -- the form the redo loop uses, where the compiler cannot keep the scalar loads (see k_exh_redo16).
template <int R, bool ALDS = false>
__device__ __forceinline__ void qsad16_block(const FastDev& d, const uint32_t* win, int pair, int brow, int bcol0, int tid,
                                             const uint32_t* alds = nullptr)
{
    constexpr int NW = R + 3;                          // 64-bit window pairs per lane and row
    const int r0 = brow * 16;
    const int NC = 2 * d.sw + 16;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bcol = bcol0 + wave;
    if (bcol >= d.nbc) return;                         // ragged last workgroup
    const int c0 = bcol * 16;
    const int lane = tid & 63;
    const int prow = lane >> 2, q = lane & 3;

    // ---- anchor block: 64 wave-uniform dwords (scalar loads -> SGPRs) ----
    const_u32* anchor = as_constant(d.prev + (long long)pair * d.plane_stride + (long long)r0 * d.pitch + c0);
    const int apitch = d.pitch >> 2;
    uint32_t A[16][4];
    if (!ALDS) {
#pragma unroll
        for (int a = 0; a < 16; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) A[a][j] = anchor[a * apitch + j];
    }

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
        if (ALDS && t < 16) {
            // volatile: keeps the read at this step (hoisted to the top, all 16 rows would be live at once)
            const u32x4_v v = *(const volatile u32x4_v*)(alds + wave * 64 + t * 4);
            A[t][0] = v.x; A[t][1] = v.y; A[t][2] = v.z; A[t][3] = v.w;
        }
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
    best = wave_min_u32(best);                           // DPP: no lane-index registers hoisted out of the redo kernel's work loop
    if (lane == 0) {
        const int idx = best & 0x1FFF;
        const int ci = idx / NC, ri = idx - ci * NC;
        int32_t* o = d.mf + (((long long)pair * d.nbr + brow) * d.nbc + bcol) * 2;
        o[0] = ci - d.sw;
        o[1] = ri - d.sw;
    }
}

template <int R>
__global__ void __launch_bounds__(384) k_exh_qsad16(FastDev d)
{
    extern __shared__ uint32_t win[];                 // [win_rows][pitch_dw]
    const int b = blockIdx.x;
    const int pair = (b / 8 / d.wg_per_pair) * 8 + (b & 7);
    if (pair >= d.pairs) return;                       // whole workgroup leaves together
    const int wg = (b >> 3) % d.wg_per_pair;
    const int brow = wg / d.wg_per_row;
    const int bcol0 = (wg - brow * d.wg_per_row) * d.nb;
    stage_window(d, win, d.cur + (long long)pair * d.plane_stride, bcol0, brow * 16, (int)threadIdx.x);
    __syncthreads();
    qsad16_block<R>(d, win, pair, brow, bcol0, (int)threadIdx.x);
}

// Redo form (hostile tiles of the elimination kernels, bbme_sea_common.h: SeaDev::redo_list): a fixed grid
// of workgroups draws items -- one block row of a listed tile each -- from a counter until the list is done.
struct RedoDev {
    const uint32_t* list;         // tile number inside its XCD's tiles << 3 | xcd
    const uint32_t* count;        // tiles listed (written by the kernel launched before this one)
    uint32_t* head;               // next item
    int tr, tile_wg_per_row, tile_wg_per_pair;      // the elimination kernel's tile grid (tiles are tr x nb blocks)
};

// ---------------------------------------------------------------------------
// Exhaustive MSE, bs = 16:  SSD = sum(A^2) + sum(B^2) - 2 sum(A.B).
//   sum(A.B)  v_dot4_u32_u8 against the SGPR anchor, same lane mapping as k_exh_qsad16; the three
//             byte-shifted copies of each window dword come from v_alignbyte_b32 (shared by the R
//             anchor rows that use them);
//   sum(B^2)  one lookup per candidate in the per-frame table built by k_sqbox16_* (every candidate
//             position is shared by up to nine macroblocks, so the table is built once per frame);
//   sum(A^2)  64 v_dot4 + a wave reduction, once per block.
// All terms are exact integers < 2^26 (bs = 16), equal to the reference's float32 sums (bbme.py:94).
// ---------------------------------------------------------------------------
// One pass over the window for the column groups K0 .. K0+KN-1 of every lane (candidates 4k .. 4k+3 of its R rows): the
// R x KN x 4 dot products, then the costs, folded into the lane's running minimum (bcost, blocal) in local scan order
// ((4k+e) outer, i inner -- the groups are visited in ascending k, so a later pass only replaces a strictly smaller cost).
// ROLL: the anchor rows are (scalar-)loaded as the window rows reach them and only the last R stay live (R x 4 scalars
// instead of 64 -- inside the work loop of k_exh_redo16 the 64 did not fit beside everything else and went to VGPRs).
template <int R, int K0, int KN, bool ROLL>
__device__ __forceinline__ void dot16_part(const FastDev& d, const uint32_t* lrow, const_u32* anchor, int apitch,
                                           SqTable sq, int tab0, bool interior, int prow, int q, int lo_r, int hi_r, int lo_c,
                                           int hi_c, int r0, int c0, uint32_t a2, uint32_t& bcost, int& blocal)
{
    constexpr int NWP = KN + 4;                        // window dwords per lane and row for these groups
    constexpr int AROWS = ROLL ? R : 16;
    uint32_t acc[R][KN][4];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int k = 0; k < KN; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][k][e] = 0;
    uint32_t A[AROWS][4];
    if (ROLL) asm volatile("" : "+s"(apitch));        // opaque per pass: merged with the other passes' loads, all 64 anchor scalars
                                                       // stay live from the first pass to the last and spill into VGPR lanes
    if (!ROLL) {
#pragma unroll
        for (int a = 0; a < 16; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) A[a][j] = anchor[a * apitch + j];
    }
#pragma unroll
    for (int t = 0; t < R + 15; ++t) {
        uint32_t w[NWP];
#pragma unroll
        for (int s = 0; s < NWP; ++s) w[s] = lrow[t * d.pitch_dw + K0 + s];
        if (ROLL && t < 16) {
#pragma unroll
            for (int j = 0; j < 4; ++j) A[t % AROWS][j] = anchor[t * apitch + j];
        }
        uint32_t sh[NWP - 1][4];                       // sh[s][e] = bytes 4(K0+s)+e .. +3 of the row
#pragma unroll
        for (int s = 0; s < NWP - 1; ++s) {
            sh[s][0] = w[s];
#pragma unroll
            for (int e = 1; e < 4; ++e) sh[s][e] = __builtin_amdgcn_alignbyte(w[s + 1], w[s], (uint32_t)e);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int a = t - i;
            if (a < 0 || a > 15) continue;
#pragma unroll
            for (int k = 0; k < KN; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[i][k][e] = __builtin_amdgcn_udot4(sh[k + j][e], A[a % AROWS][j], acc[i][k][e], false);
        }
        // Pin this row's dot products before the next row's: an empty asm that "modifies" every
        // accumulator.  Without it instruction selection linearises the unrolled body one
        // accumulator chain at a time, keeps all 18 rows of shifted dwords live (> 256 VGPRs,
        // spills) and pads the dependent v_dot4 chain with s_nops.
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int k = 0; k < KN; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(acc[i][k][e]));
        if (ROLL) __builtin_amdgcn_sched_barrier(0);   // work loop: nothing of the next window row moves up into this one
    }
    // ---- costs and the first minimum in scan order (column index outer, row index inner) ----
    // table positions as 32-bit offsets from the plane (uniform base + one VGPR; 64-bit indices cost an address pair per
    // read, and the edge branch keeps R x KN x 4 guarded reads in flight)
    if (interior) {
        // whole window inside the frame: the four candidates 4k .. 4k+3 of a row share one 16-byte table read
#pragma unroll
        for (int k = 0; k < KN; ++k) {
            uint32_t b2[R][4];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const uint4 v = *(const uint4*)(sq + (uint32_t)(tab0 + i * d.pitch + 4 * (K0 + k)));
                b2[i][0] = v.x; b2[i][1] = v.y; b2[i][2] = v.z; b2[i][3] = v.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < R; ++i) {          // local scan order: (4k+e) outer, i inner
                    const uint32_t cost = a2 + b2[i][e] - 2u * acc[i][k][e];
                    if (cost < bcost) { bcost = cost; blocal = (4 * (K0 + k) + e) * R + i; }
                }
        }
    } else {
#pragma unroll
        for (int k = 0; k < KN; ++k) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < R; ++i) {          // local scan order: (4k+e) outer, i inner
                    const int ri = prow * R + i, ci = q * 4 * R + 4 * (K0 + k) + e;
                    if (ri >= lo_r && ri <= hi_r && ci >= lo_c && ci <= hi_c) {
                        const uint32_t b2 = sq[(uint32_t)((r0 - d.sw + ri) * d.pitch + (c0 - d.sw + ci))];
                        const uint32_t cost = a2 + b2 - 2u * acc[i][k][e];
                        if (cost < bcost) { bcost = cost; blocal = (4 * (K0 + k) + e) * R + i; }
                    }
                }
            if (ROLL) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// KS = column groups per pass over the window.  KS = R is the one-pass form (k_exh_dot16: a kernel of its own may take
// 166 VGPRs at R = 5); the work-loop kernel k_exh_redo16 runs R = 3 .. 5 in passes of two groups -- R x 2 x 4 accumulators
// instead of R x R x 4 (100 at R = 5, which spilled 215 VGPRs to scratch inside the 128 that four waves per SIMD allow,
// VERDICT r3 #2) -- at the price of the window's dwords and their byte shifts being fetched once per pass (v_dot4 work, 94 %
// of the body, is unchanged).
template <int R, int KS = R>
__device__ __forceinline__ void dot16_block(const FastDev& d, const uint32_t* win, int pair, int brow, int bcol0, int tid)
{
    const int r0 = brow * 16;
    const int NC = 2 * d.sw + 16;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bcol = bcol0 + wave;
    if (bcol >= d.nbc) return;
    const int c0 = bcol * 16;
    const int lane = tid & 63;
    const int prow = lane >> 2, q = lane & 3;

    const uint8_t* aptr = d.prev + (long long)pair * d.plane_stride + (long long)r0 * d.pitch + c0;
    // wave-uniform by construction (pair, row, wave); said explicitly so that the anchor reads stay scalar loads
    const unsigned long long abits = (unsigned long long)aptr;
    const_u32* anchor = as_constant((const void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(abits >> 32)) << 32) |
                                                  (unsigned)__builtin_amdgcn_readfirstlane((int)abits)));
    const int apitch = d.pitch >> 2;
    // sum of squares of the anchor: lane l takes dword l of the block
    uint32_t a2;
    {
        const uint32_t mine = *(const uint32_t*)(aptr + (long long)(lane >> 2) * d.pitch + (lane & 3) * 4);
        a2 = wave_sum_u32(__builtin_amdgcn_udot4(mine, mine, 0u, false));      // DPP: no lane-index registers (ds_bpermute
    }                                                                           // addresses were hoisted out of the work loop and spilled)

    const uint32_t* lrow = win + (prow * R) * d.pitch_dw + wave * 4 + q * R;
    const int lo_r = max(0, d.sw - r0), hi_r = min(NC - 1, d.H - 16 - r0 + d.sw);
    const int lo_c = max(0, d.sw - c0), hi_c = min(NC - 1, d.W - 16 - c0 + d.sw);
    const SqTable sq = sq_table(d.sqbox + (long long)pair * d.sqbox_stride, d.H, d.pitch);
    const int tab0 = (r0 - d.sw + prow * R) * d.pitch + (c0 - d.sw + q * 4 * R);       // >= 0 for interior blocks (only they use it)
    uint32_t bcost = 0xFFFFFFFFu;
    int blocal = 0;
    const bool interior = __builtin_amdgcn_readfirstlane(lo_r == 0 && lo_c == 0 && hi_r == 16 * R - 1 && hi_c == 16 * R - 1);
#define DOT16_PART(K0) \
    if constexpr ((K0) < R) dot16_part<R, (K0), ((K0) + KS <= R ? KS : R - (K0)), (KS < R)>(d, lrow, anchor, apitch, sq, tab0, interior, prow, q, \
                                                                                         lo_r, hi_r, lo_c, hi_c, r0, c0, a2, bcost, blocal)
    DOT16_PART(0);
    DOT16_PART(KS);
    DOT16_PART(2 * KS);
    DOT16_PART(3 * KS);
    DOT16_PART(4 * KS);
#undef DOT16_PART
    // the smallest (cost << 13 | scan index) of the wave in two 32-bit steps: the smallest cost, then the smallest scan index
    // among the lanes that hold it (bbme.py:171: the first minimum in scan order)
    const uint32_t cmin = wave_min_u32(bcost);
    uint32_t myidx = 0xFFFFFFFFu;
    if (bcost == cmin && bcost != 0xFFFFFFFFu) {
        const int ce = blocal / R, i = blocal - ce * R;
        myidx = (uint32_t)((q * 4 * R + ce) * NC + prow * R + i);
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

template <int R>
__global__ void __launch_bounds__(384) k_exh_dot16(FastDev d)
{
    extern __shared__ uint32_t win[];
    const int b = blockIdx.x;
    const int pair = (b / 8 / d.wg_per_pair) * 8 + (b & 7);
    if (pair >= d.pairs) return;
    const int wg = (b >> 3) % d.wg_per_pair;
    const int brow = wg / d.wg_per_row;
    const int bcol0 = (wg - brow * d.wg_per_row) * d.nb;
    stage_window(d, win, d.cur + (long long)pair * d.plane_stride, bcol0, brow * 16, (int)threadIdx.x);
    __syncthreads();
    dot16_block<R>(d, win, pair, brow, bcol0, (int)threadIdx.x);
}

#ifndef REDO_KS5
// column groups per pass of the R = 5 MSE body: 2 -> 128 VGPRs, 4 waves per SIMD; 1 -> 96 VGPRs, 5 waves (both: no spill, no scratch).
// Same box, 128 pairs of 1080p noise, sw 32: 6.29 k pairs/s with 2, 6.01 k with 1 (the fifth wave does not pay for three more
// passes over the window), 4.37 k for round 3's one-pass body (215 VGPRs spilled to 192 bytes of scratch per lane).
#define REDO_KS5 2
#endif
// The work-loop form of the two bodies above.  The anchors stay scalar here too (constant address space, see
// as_constant): with plain global loads the compiler moved them to VGPRs inside the loop (86-128 per wave), round 2
// staged them in LDS instead (82 / 117 VGPRs: 5 / 4 waves per SIMD).
template <int R, bool MSE>
__global__ void __launch_bounds__(1024, (MSE && (R == 4 || (R == 5 && REDO_KS5 == 1)) ? 5 : 4)) k_exh_redo16(FastDev d, RedoDev r)
{
    extern __shared__ uint32_t win[];
    __shared__ uint32_t item_s;
    const uint32_t total = *r.count * (uint32_t)r.tr;
    if (total == 0) return;                                // friendly content: nothing was listed, no atomic is spent
    for (;;) {
        if (threadIdx.x == 0) item_s = atomicAdd(r.head, 1u);
        __syncthreads();
        const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item_s);
        if (item >= total) break;
        const uint32_t ent = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.list[item / (uint32_t)r.tr]);
        const int row_in_tile = (int)(item % (uint32_t)r.tr), xcd = (int)(ent & 7u), t = (int)(ent >> 3);
        const int lp = t / r.tile_wg_per_pair, wg = t - lp * r.tile_wg_per_pair;
        const int trow = wg / r.tile_wg_per_row, bcol0 = (wg - trow * r.tile_wg_per_row) * d.nb;
        const int pair = lp * 8 + xcd, brow = trow * r.tr + row_in_tile;
        const bool ok = brow < d.nbr && pair < d.pairs;                                     // ragged last tile row
        // MSE: everything derived from the thread index is recomputed per item from an opaque copy -- hoisted out of the
        // work loop those values (window offsets, table addresses) cost ~20 VGPRs in the MSE body (112 -> 91: 4 -> 5 waves
        // per SIMD, +2.8 % on noise).  The MAE body is better off with the hoisting (72 VGPRs either way 7 waves; 9 scalar
        // spills and -1.8 % with the opaque copy).
        int tid = (int)threadIdx.x;
        if (MSE || R == 5) asm volatile("" : "+v"(tid));     // (MAE at R = 5: 128 VGPRs + 2 spilled without it)
        if (ok) stage_window(d, win, d.cur + (long long)pair * d.plane_stride, bcol0, brow * 16, tid);
        __syncthreads();
        if (ok) {
            if (MSE) dot16_block<R, (R >= 5 ? REDO_KS5 : R >= 3 ? 2 : R)>(d, win, pair, brow, bcol0, tid);
            else qsad16_block<R>(d, win, pair, brow, bcol0, tid);
        }
        __syncthreads();                                   // the next item restages `win` and redraws item_s
    }
}

// 16x16 box sums of squares of one plane stack in ONE pass (5.5 bytes of HBM traffic per pixel
// instead of the ~15 of a rows pass + a columns pass through a uint32 scratch plane).
// Thread -> four adjacent columns x .. x+3 (x % 4 == 0) of a chunk of SQ_CHUNK output rows:
//   h(row) = the four horizontal 16-byte sums of squares of `row`: five aligned dwords, three
//            v_alignbyte copies of each, sixteen v_dot4(b, b);
//   S(y)   = S(y-1) + h(y+15) - h(y-1), the last 16 h vectors held in a register ring (the row
//            loop is unrolled in groups of 16 so that the ring indices are compile-time).
// Each chunk re-walks 15 warm-up rows (1 byte per pixel, cheap next to the 4-byte outputs).
#ifndef SQ_CHUNK_ROWS
#define SQ_CHUNK_ROWS 64
#endif
constexpr int SQ_CHUNK = SQ_CHUNK_ROWS;

// SGN: the squares of (b - 128) instead of b -- the table of the MFMA search (bbme_mfma.hip), whose int8 operands are
// the frames' bytes with the top bit flipped
template <bool SGN>
__device__ __forceinline__ u32x4_v sq_hsum4(const uint8_t* row, int x, int pitch)
{
    const uint32_t* p = (const uint32_t*)row;
    uint32_t w[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) w[j] = (x + 4 * j < pitch) ? p[j] : 0u;     // last dword may lie past the pitch
    // SGN: sum (b - 128)^2 = sum b^2 - 256 sum b + 16 * 16384 per 16 bytes, both sums by v_dot4_u32_u8 (the signed dot
    // product, v_dot4c_i32_i8, ran this kernel at half the speed: 1.84 against 0.87 ms per 2049 frames of 720x480)
    uint32_t r[4], sb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r[e] = SGN ? 16u * 16384u : 0u;
        sb[e] = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t v = e == 0 ? w[j] : __builtin_amdgcn_alignbyte(w[j + 1], w[j], (uint32_t)e);
            r[e] = __builtin_amdgcn_udot4(v, v, r[e], false);
            if (SGN) sb[e] = __builtin_amdgcn_udot4(v, 0x01010101u, sb[e], false);
        }
        if (SGN) r[e] -= 256u * sb[e];
    }
    u32x4_v out;
    out.x = r[0]; out.y = r[1]; out.z = r[2]; out.w = r[3];          // columns beyond W-16 are never read
    return out;
}

// Round 3: no register ring.  Round 2 kept the last 16 row vectors h in 64 VGPRs (152 in all: 3 waves per SIMD, and a
// write-bound kernel at 3.4 TB/s with nobody to cover its latencies); the row that leaves the window, h(y - 1), is now
// recomputed from a second read of that source row (16 rows back in the same thread's walk: L2 hits) -- twice the vector
// work of a kernel that was 26 % VALU-busy, a quarter of the registers.
// (Round 4 tried the ring in LDS -- 16 x 16 bytes per thread, 64 KiB per workgroup, minimum traffic: 2 waves per SIMD are too
// few for a streaming kernel, exhaustive MSE 328 k -> 291 k pairs/s, same box.)
template <bool SGN>
__global__ void __launch_bounds__(256) k_sqbox16(const uint8_t* src, long long src_stride, int H, int W, int pitch,
                                                 uint32_t* dst, long long dst_stride)
{
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * SQ_CHUNK;
    if (x > W - 16 || y0 > H - 16) return;
    const uint8_t* p = src + (long long)blockIdx.z * src_stride + (long long)y0 * pitch + x;
    uint32_t* o = dst + (long long)blockIdx.z * dst_stride + (long long)y0 * pitch + x;
    u32x4_v s = { 0, 0, 0, 0 };
#pragma unroll 5
    for (int r = 0; r < 15; ++r) s += sq_hsum4<SGN>(p + (long long)r * pitch, x, pitch);        // rows y0 .. y0+14 <= H-2
    const int last = min(SQ_CHUNK - 1, H - 16 - y0);     // last output row of this chunk
#pragma unroll 2
    for (int k = 0; k <= last; ++k) {
        const u32x4_v hn = sq_hsum4<SGN>(p + (long long)(k + 15) * pitch, x, pitch);            // row y0+k+15 <= H-1 enters
        const u32x4_v ho = sq_hsum4<SGN>(p + (long long)k * pitch, x, pitch);                   // row y0+k leaves behind this output
        s += hn;
        // non-temporal: the 4-byte-per-pixel output stream would otherwise push the source rows out of the L2 before
        // their second read 16 rows later (same box: 315.2 k -> 319.3 k pairs/s exhaustive MSE, 30.45 k -> 30.8 k at 1080p)
        __builtin_nontemporal_store(s, (u32x4_v*)(o + (long long)k * pitch));
        s -= ho;
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
    if (job.procedure != GME_SEARCH_EXHAUSTIVE || job.bs != 16) return GME_OK;
    const bool mse = job.pnorm == GME_NORM_MSE;
    if (mse && (job.sqbox_cur == nullptr || getenv("GME_FORCE_GENERIC"))) return GME_OK;
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
    d.sqbox = job.sqbox_cur; d.sqbox_stride = job.sqbox_stride;
    // waves per workgroup: prefer an exact divisor of the block-row length
    int nb = 4;
    for (int cand : {5, 4, 6, 3}) if (nbc % cand == 0) { nb = cand; break; }
    if (nbc < nb) nb = nbc;
    d.nb = nb;
    d.wg_per_row = (nbc + nb - 1) / nb;
    d.wg_per_pair = d.wg_per_row * nbr;
    d.win_rows = 16 * R + 15;                          // rows a lane may touch: prow*R + t, t < R+15
    // bytes a lane may touch in a row: wave*16 + q*4R + 4*(R+3) + 8  (last 64-bit pair)
    const int need_dw = (nb - 1) * 4 + 3 * R + (R + 2) + 2;      // also covers k_exh_dot16's R + 4 dwords
    d.pitch_dw = pick_pitch_dw(need_dw, R);
    const size_t lds = (size_t)d.win_rows * d.pitch_dw * 4;
    const long long groups = (long long)((job.pairs + 7) / 8) * 8 * d.wg_per_pair;
    GME_REQUIRE(groups < (1ll << 31), GME_ERR_ARG, "too many workgroups in one launch");
    const dim3 grid((unsigned)groups), block(64 * nb);
    plan_note(ctx, 0, "%s<%d> 1x%d blocks per workgroup grid %lld", mse ? "k_exh_dot16" : "k_exh_qsad16", R, nb, groups);
    if (mse) {
        switch (R) {
        case 1: hipLaunchKernelGGL(k_exh_dot16<1>, grid, block, lds, ctx->stream, d); break;
        case 2: hipLaunchKernelGGL(k_exh_dot16<2>, grid, block, lds, ctx->stream, d); break;
        case 3: hipLaunchKernelGGL(k_exh_dot16<3>, grid, block, lds, ctx->stream, d); break;
        case 4: hipLaunchKernelGGL(k_exh_dot16<4>, grid, block, lds, ctx->stream, d); break;
        default: hipLaunchKernelGGL(k_exh_dot16<5>, grid, block, lds, ctx->stream, d); break;
        }
        GME_HIP_TRY(hipGetLastError());
        *handled = true;
        return GME_OK;
    }
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

// Brute-force search of the tiles an elimination kernel listed (launched right behind it on the same stream).
// nb = the tile's width in blocks; the grid is fixed, workgroups that find the list empty leave at once.
int launch_exh_redo(gme_ctx* ctx, const BbmeJob& job, int R, int tr, int tc, int tile_wg_per_row, int tile_wg_per_pair,
                    const uint32_t* list, const uint32_t* count, uint32_t* head)
{
    const bool mse = job.pnorm == GME_NORM_MSE;
    FastDev d;
    d.prev = job.prev; d.cur = job.cur; d.plane_stride = job.plane_stride;
    d.pairs = job.pairs; d.H = job.H; d.W = job.W; d.pitch = job.pitch; d.sw = job.sw;
    d.nbr = job.H / 16; d.nbc = job.W / 16; d.mf = job.mf;
    d.sqbox = job.sqbox_cur; d.sqbox_stride = job.sqbox_stride;
    d.nb = tc;
    d.wg_per_row = tile_wg_per_row; d.wg_per_pair = tile_wg_per_pair * tr;     // unused by the redo form
    d.win_rows = 16 * R + 15;
    const int need_dw = (tc - 1) * 4 + 3 * R + (R + 2) + 2;
    d.pitch_dw = pick_pitch_dw(need_dw, R);
    const size_t lds = (size_t)d.win_rows * d.pitch_dw * 4;
    RedoDev r;
    r.list = list; r.count = count; r.head = head;
    r.tr = tr; r.tile_wg_per_row = tile_wg_per_row; r.tile_wg_per_pair = tile_wg_per_pair;
    int per_cu = 32 / tc;                                                   // resident workgroups: 32 wave slots ...
    const int by_lds = (int)((160 * 1024) / (lds + 1024));                   // ... and LDS
    if (per_cu > by_lds) per_cu = by_lds;
    if (per_cu < 1) per_cu = 1;
    // a full grid costs ~30 us to dispatch even when every workgroup finds the list empty: never launch more
    // workgroups than the list can hold items (single-pair calls: a few hundred)
    long long groups = (long long)per_cu * ctx->prop.multiProcessorCount;
    const long long max_items = (long long)job.pairs * tile_wg_per_pair * tr;
    if (groups > max_items) groups = max_items;
    const dim3 grid((unsigned)groups), block(64 * tc);
#define REDO_LAUNCH(RR) do { if (mse) hipLaunchKernelGGL((k_exh_redo16<RR, true>), grid, block, lds, ctx->stream, d, r); \
                             else hipLaunchKernelGGL((k_exh_redo16<RR, false>), grid, block, lds, ctx->stream, d, r); } while (0)
    switch (R) {
    case 1: REDO_LAUNCH(1); break;
    case 2: REDO_LAUNCH(2); break;
    case 3: REDO_LAUNCH(3); break;
    case 4: REDO_LAUNCH(4); break;
    default: REDO_LAUNCH(5); break;
    }
#undef REDO_LAUNCH
    GME_HIP_TRY(hipGetLastError());
    return GME_OK;
}

// 16x16 box sums of squares for `count` planes: `out` is a uint32 stack with the frame's pitch
// (elements) and `stride` elements between planes.
int launch_sqbox16(gme_ctx* ctx, const uint8_t* src, long long src_stride, int count, int H, int W, int pitch,
                   uint32_t* out, long long stride, bool sgn)
{
    if (count == 0 || H < 16 || W < 16) return GME_OK;
    const int step = max_grid_planes();
    for (int first = 0; first < count; first += step) {
        const int n = count - first < step ? count - first : step;
        const int xq = (W - 16) / 4 + 1;                            // column quads that hold a valid position
        const dim3 g((xq + 63) / 64, ((H - 15 + SQ_CHUNK - 1) / SQ_CHUNK + 3) / 4, n);
        if (sgn) hipLaunchKernelGGL(k_sqbox16<true>, g, dim3(256), 0, ctx->stream, src + first * src_stride, src_stride, H, W, pitch,
                                    out + first * stride, stride);
        else hipLaunchKernelGGL(k_sqbox16<false>, g, dim3(256), 0, ctx->stream, src + first * src_stride, src_stride, H, W, pitch,
                                out + first * stride, stride);
    }
    GME_HIP_TRY(hipGetLastError());
    return GME_OK;
}

int bbme_aux_kind(int bs, int sw, int procedure, int pnorm)
{
    if (procedure != GME_SEARCH_EXHAUSTIVE || bs != 16 || pnorm != GME_NORM_MSE) return 0;
    if (sw < 0 || sw % 4 != 0) return 0;
    const int NC = 2 * sw + 16;
    if (!((NC + 15) / 16 <= 5 && NC * NC <= 8192 && !getenv("GME_FORCE_GENERIC"))) return 0;
    return bbme_mfma_wanted(sw) ? 2 : 1;
}

int launch_aux_table(gme_ctx* ctx, int kind, const uint8_t* src, long long src_stride, int count, int H, int W,
                     int pitch, uint32_t* out, long long stride)
{
    if (kind == 1 || kind == 2) return launch_sqbox16(ctx, src, src_stride, count, H, W, pitch, out, stride, kind == 2);
    return GME_OK;
}
