// Walk searches (diamond, three-step, 2-D log) specialised for the two geometries the
// global-motion pipeline runs all the time (motion.py:27-29,224-229):
//
//   k_walk16<PNORM> (diamond), k_walk16s<PNORM, PROC, FITS> (three-step, 2-D log)  bs = 16: one wavefront per macroblock,
//                     8 blocks per wave one after the other (the next block's anchors -- and, for the diamond and the
//                     three-step search, its first window -- are fetched while the current one is walked, two register
//                     sets used in turn; the wave steps its block's grid position instead of dividing).  The wave is cut
//                     into 8 groups of 8 lanes; a group evaluates one candidate per round, each lane owning
//                     two 16-byte rows of the block, s and s + 8 (ten dwords of the per-wave LDS window of
//                     `cur`, 8 v_alignbyte_b32, 8 v_sad_u8 or 16 v_dot4_u32_u8 against its 8
//                     anchor dwords kept in VGPRs), followed by a 3-step DPP reduction inside
//                     the group.  Up to 8 candidates cost one round; the centre of a pattern is
//                     the previous winner, whose cost is carried instead of recomputed (diamond: every round; three-step:
//                     second step, and third when the first did not move; 2-D log: the closing ring).  Under MSE a block's
//                     own sum of a^2 is a constant of every comparison the walk makes and is replaced by one (MSE_BIAS).
//                     The window is staged branch-free through a buffer resource (out-of-plane reads return 0).
//                     k_walk16 is the diamond search on its own (the GME levels 1-2): rounds pick
//                     the winner in the vector unit (PATTERN_MIN), 41 vector + ~35 scalar instructions
//                     per round of 8 candidates under MSE.  Three-step and 2-D log are instances of their own
//                     (PROC): every group derives its candidate from its index, validity is a per-lane test, the
//                     round's box is the pattern's span clipped to the frame (EVALV) and the winner comes
//                     from the same DPP minimum -- no wave-uniform candidate arrays, no scalar spills.  Their LDS
//                     window is 48 rows x 64 bytes (Win<true>): the widest round of both searches at sw <= 16 (span 32)
//                     fits (FITS: no code that reads global memory directly); wider searches take the FITS = false instance.
//   k_dense2<PNORM>   bs = 2, diamond: one lane per 2x2 block (the dense first estimate on the
//                     coarsest pyramid level, 5400 blocks per 720x480 pair).
//
// Candidate order, clamping/validity rules and tie-breaking are those of
// bbme.py:182-534 (see bbme_kernels.hip for the generic form and the quirk list).
#include "gme_internal.h"

namespace {

struct WalkDev {
    const uint8_t* prev;
    const uint8_t* cur;
    long long plane_stride;
    int pairs, H, W, pitch, sw, procedure;
    int nbr, nbc, bpw;
    int st1, st2, st3;        // three-step: int((2 sw + bs) / 3), / 5, / 10 (bbme.py:211-213)
    int32_t* mf;
    int* status;
};

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint16_t u16_u __attribute__((aligned(1)));

constexpr unsigned INF32 = 0xFFFFFFFFu;
constexpr unsigned MSE_BIAS = 1u << 24;        // stands in for a block's sum of a^2 in the walk kernels' MSE costs (see walk_block)

__device__ __forceinline__ int clamp_ref(int v, int hi)   // min(max(v, 0), hi), bbme.py:503-504
{
    const int t = v > 0 ? v : 0;
    return t < hi ? t : hi;
}

// clamp_ref for per-lane values as one instruction (v_med3_i32 of v, 0, hi; hi >= 0 on every path that gets here:
// bbme_check_args refuses diamond searches on frames without room for a block)
__device__ __forceinline__ int clamp_med3(int v, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "s"(hi));
    return r;
}

// sum over the 8 lanes of a group (lanes 8g .. 8g+7) with DPP moves: xor 1, xor 2 inside
// the quad, then the half-row mirror brings in the other quad
__device__ __forceinline__ unsigned group8_sum(unsigned v)
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);   // row_half_mirror
    return v;
}

// cost of up to 8 candidates (one per 8-lane group); positions are wave-uniform arrays.
// Returns the group's candidate cost in every lane of the group (INF32 if !valid).
template <int PNORM>
__device__ __forceinline__ unsigned group_eval(const uint32_t (&a)[8], unsigned aa, const uint8_t* cur, int pitch,
                                               int rr, int cc, bool valid, int lrow)
{
    unsigned part = 0;
    if (valid) {
        // dword-aligned loads + v_alignbyte: byte-misaligned 16-byte loads are legal on gfx950
        // but crawl through the texture addresser (measured 4x slower end to end)
        const uint8_t* p = cur + (long long)(rr + lrow) * pitch + (cc & ~3);
        const uint32_t sh = (uint32_t)cc & 3u;
        const u32x4_a4 l0 = *(const u32x4_a4*)p, l1 = *(const u32x4_a4*)(p + 8 * pitch);
        const uint32_t t0 = *(const uint32_t*)(p + 16), t1 = *(const uint32_t*)(p + 8 * pitch + 16);
        const uint32_t b[8] = { __builtin_amdgcn_alignbyte(l0.y, l0.x, sh), __builtin_amdgcn_alignbyte(l0.z, l0.y, sh),
                                __builtin_amdgcn_alignbyte(l0.w, l0.z, sh), __builtin_amdgcn_alignbyte(t0, l0.w, sh),
                                __builtin_amdgcn_alignbyte(l1.y, l1.x, sh), __builtin_amdgcn_alignbyte(l1.z, l1.y, sh),
                                __builtin_amdgcn_alignbyte(l1.w, l1.z, sh), __builtin_amdgcn_alignbyte(t1, l1.w, sh) };
        if (PNORM == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) part = __builtin_amdgcn_sad_u8(a[j], b[j], part);
        } else {
            unsigned bb = 0, ab = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bb = __builtin_amdgcn_udot4(b[j], b[j], bb, false);
                ab = __builtin_amdgcn_udot4(a[j], b[j], ab, false);
            }
            part = aa + bb - 2u * ab;          // this lane's 32 pixels of b^2 - 2ab plus its share of MSE_BIAS
        }
    }
    part = group8_sum(part);
    return valid ? part : INF32;
}

// Search-window cache: WIN_ROWS x WIN_DW dwords of `cur` per wave in LDS (row pitch WIN_PITCH is
// odd, so the 8 lanes of a group -- on 8 consecutive rows, in each of their two reads -- hit 8 different banks).  A candidate block
// (rr, cc) can be served from it when 0 <= rr - wr0 <= WIN_ROWS - 16 and 0 <= cc - wc0 <= WIN_SPAN.
constexpr int WIN_ROWS = 40, WIN_DW = 12, WIN_PITCH = 13, WIN_SPAN = 4 * (WIN_DW - 5) + 3;
constexpr int WIN_ALLOC = 43 * WIN_PITCH;      // staging moves 128 sixteen-byte segments (2 per lane, no idle lanes): 42 2/3 rows

// Two window geometries.  Win<false> (the constants above) is the diamond walk's: patterns reach 2 pixels, the window
// follows the walk.  Win<true> is the one of three-step and 2-D log (round 3): 48 rows x 64 bytes (192 segments, 3 per
// lane, pitch 17 dwords: odd, so the 8 lanes of a group still hit 8 banks) hold a pattern of span 32 -- the first step
// of both searches at sw = 16 -- so that round is served from the LDS as well instead of reading global memory through
// unaligned loads, and the later, narrower rounds usually fall inside the same window: one staging per block.
// which FITS searches fetch their first window ahead (bit PROC): it costs 12 VGPRs per prefetch buffer
#ifndef WALK_FITS_PREFETCH
#define WALK_FITS_PREFETCH (1 << GME_SEARCH_THREESTEP)
#endif
#ifndef WALK_SMALLWIN
constexpr bool WALK_BIG = true;
#else
constexpr bool WALK_BIG = false;
#endif
template <bool BIG> struct Win {
    static constexpr int ROWS = BIG ? 48 : WIN_ROWS, DW = BIG ? 16 : WIN_DW, PITCH = BIG ? 17 : WIN_PITCH;
    static constexpr int SPAN = 4 * (DW - 5) + 3;
    static constexpr int SEGS_ROW = BIG ? 4 : 3, SEGS_LANE = BIG ? 3 : 2;
    static constexpr int ALLOC = BIG ? 48 * 17 : WIN_ALLOC;
};

// Window staging in two halves, so that the loads of a block's first window can be in flight while the
// wave still walks the block before it: 3 sixteen-byte segments per row, 2 segments per lane.
// buffer resource over one H x pitch plane plus `slack` bytes (pointer and size wave-uniform)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const uint8_t* plane, int bytes)
{
    const uint64_t base = (uint64_t)plane;
    return __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((uint64_t)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                (uint32_t)__builtin_amdgcn_readfirstlane((int)base)),
        (short)0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

template <bool BIG = false>
__device__ __forceinline__ void window_load(uint32_t (&v)[Win<BIG>::SEGS_LANE][4], const uint8_t* cur, int pitch, int H, int wr0, int wc0,
                                            int lane)
{
    // Branch-free through a buffer resource over the plane: whatever lies outside [0, H * pitch) -- rows above or
    // below the frame -- reads as 0.  The window never starts left of the frame (wc0 >= 0: a 16-byte load whose
    // first dword lies before the plane is dropped as a whole, valid bytes included); columns right of a
    // row wrap into the next row, which is harmless: candidates are clamped into the frame, so window cells
    // outside it are never part of a cost.  Without branches the loads of the NEXT block's window stay in flight
    // (no s_waitcnt between them and the walk of the current block).
    // The range is 12 bytes longer than the plane so that a 16-byte segment starting in the last 12 bytes of the
    // last row is not dropped as a whole; every plane has a guard row behind it (plane_alloc, gme_bbme_u8).
    const __amdgpu_buffer_rsrc_t rs = plane_rsrc(cur, H * pitch + 12);
#pragma unroll
    for (int it = 0; it < Win<BIG>::SEGS_LANE; ++it) {
        const int seg = lane + 64 * it;
        const int row = seg / Win<BIG>::SEGS_ROW, s4 = seg - row * Win<BIG>::SEGS_ROW;
        const int off = __mul24(wr0 + row, pitch) + wc0 + 16 * s4;           // negative = far out of range as unsigned
        const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        v[it][0] = t.x; v[it][1] = t.y; v[it][2] = t.z; v[it][3] = t.w;
    }
}

template <bool BIG = false>
__device__ __forceinline__ void window_store(uint32_t* lds, const uint32_t (&v)[Win<BIG>::SEGS_LANE][4], int lane)
{
#pragma unroll
    for (int it = 0; it < Win<BIG>::SEGS_LANE; ++it) {
        const int seg = lane + 64 * it;                 // small: 128 segments = the 40 window rows + 8 spare segments (WIN_ALLOC)
        const int row = seg / Win<BIG>::SEGS_ROW, s4 = seg - row * Win<BIG>::SEGS_ROW;
        uint32_t* o = lds + row * Win<BIG>::PITCH + 4 * s4;
        o[0] = v[it][0]; o[1] = v[it][1]; o[2] = v[it][2]; o[3] = v[it][3];
    }
    __builtin_amdgcn_wave_barrier();                    // LDS ops of one wave complete in order
}

template <bool BIG = false>
__device__ __forceinline__ void stage_walk_window(uint32_t* lds, const uint8_t* cur, int pitch, int H, int wr0,
                                                  int wc0, int lane)
{
    uint32_t v[Win<BIG>::SEGS_LANE][4];
    window_load<BIG>(v, cur, pitch, H, wr0, wc0, lane);
    window_store<BIG>(lds, v, lane);
}

typedef __attribute__((address_space(3))) const uint32_t lds_u32;

__device__ __forceinline__ int lds_address(const uint32_t* p)       // byte address of a __shared__ object inside the LDS
{
    return (int)(uint32_t)(uintptr_t)(lds_u32*)p;
}

// cost of the candidate whose rows lrow and lrow + 8 for this lane start at LDS byte address `addr` and 8 rows below it
// (shift sh inside the dword)
template <int PNORM, int PITCH = WIN_PITCH>
__device__ __forceinline__ unsigned group_eval_at(const uint32_t (&a)[8], unsigned aa, int addr, uint32_t sh, bool valid)
{
    unsigned part = 0;
    if (valid) {
        lds_u32* p = (lds_u32*)(uint32_t)addr;
        uint32_t l0[5], l1[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) { l0[j] = p[j]; l1[j] = p[8 * PITCH + j]; }
        uint32_t b[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b[j] = __builtin_amdgcn_alignbyte(l0[j + 1], l0[j], sh);
            b[4 + j] = __builtin_amdgcn_alignbyte(l1[j + 1], l1[j], sh);
        }
        if (PNORM == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) part = __builtin_amdgcn_sad_u8(a[j], b[j], part);
        } else {
            unsigned bb = aa, ab = 0;                      // the lane's share of MSE_BIAS rides in the b^2 chain: no separate add
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bb = __builtin_amdgcn_udot4(b[j], b[j], bb, false);
                ab = __builtin_amdgcn_udot4(a[j], b[j], ab, false);
            }
            // bb - 2 ab in one op (ab < 2^23).  The s_nop covers the dot4 -> VALU read wait states that the
            // compiler's hazard pass only inserts for consumers it can see (it gave the same 3 to its own v_lshlrev here)
            asm("s_nop 2\n\tv_mad_i32_i24 %0, %1, -2, %2" : "=v"(part) : "v"(ab), "v"(bb));
        }
    }
    part = group8_sum(part);
    return valid ? part : INF32;
}

static_assert(4 * WIN_PITCH == 52, "row pitch in bytes is spelled out in the asm below");

// byte offset of candidate (rr, cc)'s rows lrow, lrow + 1 in the window.  Window origin columns are multiples of 4
// (wc0 = ... & ~3), so the byte shift is cc & 3 and the dword column (cc & ~3) - wc0; everything wave-uniform
// (window origin: sbase = -wr0 * row bytes - wc0) folds into one scalar: mad24 + and + add
template <int PITCH = WIN_PITCH>
__device__ __forceinline__ int window_offset(int sbase, int rr, int cc, int lrow52)
{
    int rowoff;                                              // valid candidates have 0 <= rr < 2^24; asm keeps the compiler
    if (PITCH == WIN_PITCH)                                  // from re-deriving a v_mul_lo_u32
        asm("v_mad_u32_u24 %0, %1, 52, %2" : "=v"(rowoff) : "v"(rr), "v"(lrow52));
    else                                                     // 68 is no inline constant: the row bytes come in an SGPR
        asm("v_mad_u32_u24 %0, %1, %3, %2" : "=v"(rowoff) : "v"(rr), "v"(lrow52), "s"(4 * PITCH));
    return rowoff + ((cc & ~3) + sbase);
}

template <int PNORM, bool BIG = false>
__device__ __forceinline__ unsigned group_eval_lds(const uint32_t (&a)[8], unsigned aa, const uint32_t* lds, int sbase,
                                                   int rr, int cc, bool valid, int lrow)
{
    constexpr int P = Win<BIG>::PITCH;
    return group_eval_at<PNORM, P>(a, aa, window_offset<P>(sbase + lds_address(lds), rr, cc, lrow * (4 * P)), (uint32_t)cc & 3u, valid);
}

// cost of ONE candidate by all 64 lanes together, wave-uniform result: lane l takes dword l % 4 of block row l / 4 (two
// window dwords, one v_alignbyte, one v_sad_u8 or two v_dot4), 7 DPP adds leave the sum in lane 63.  `origin` = LDS
// byte address of window cell (row 0, column 0) in frame coordinates (window base - wr0 * row bytes - wc0).  A third of
// the instructions of a group round that would keep 7 of the 8 groups idle: the diamond's first centre, the ninth
// candidate of a three-step step and of the 2-D log's last ring.
template <int PNORM, int PITCH>
__device__ __forceinline__ unsigned wave_eval_lds(uint32_t mine, int origin, int rr, int cc, int lane)
{
    lds_u32* p = (lds_u32*)(uint32_t)((lane >> 2) * (4 * PITCH) + 4 * (lane & 3) + (rr * (4 * PITCH) + (cc & ~3) + origin));
    const uint32_t b = __builtin_amdgcn_alignbyte(p[1], p[0], (uint32_t)cc & 3u);
    unsigned part;
    if (PNORM == 0) {
        part = __builtin_amdgcn_sad_u8(mine, b, 0u);
    } else {
        const unsigned bb = __builtin_amdgcn_udot4(b, b, MSE_BIAS / 64, false);
        const unsigned ab = __builtin_amdgcn_udot4(mine, b, 0u, false);
        asm("s_nop 2\n\tv_mad_i32_i24 %0, %1, -2, %2" : "=v"(part) : "v"(ab), "v"(bb));
    }
    part = group8_sum(part);
    part += (unsigned)__builtin_amdgcn_update_dpp(0, (int)part, 0x128, 0xF, 0xF, false);     // row_ror 8: 16-lane rows
    part += (unsigned)__builtin_amdgcn_update_dpp(0, (int)part, 0x142, 0xA, 0xF, false);     // row_bcast15 into rows 1, 3
    part += (unsigned)__builtin_amdgcn_update_dpp(0, (int)part, 0x143, 0xC, 0xF, false);     // row_bcast31 into rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane((int)part, 63);
}

// Four candidates, one per 16-lane row, one block row per lane (the diamond's small pattern: a group round would leave
// half of the wave idle).  A lane's two block rows are lrow and lrow + 8 (walk_block), so lane r of a 16-lane row already
// holds block row r: in a[0..3] when r < 8, in a[4..7] otherwise -- `arow` is that choice, no data moves.  Five window
// dwords, 4 v_alignbyte, 4 v_sad_u8 or 8 v_dot4, 4 DPP adds leave the candidate's cost in every lane of its row.
template <int PNORM>
__device__ __forceinline__ unsigned row16_eval_at(const uint32_t (&arow)[4], int addr, uint32_t sh)
{
    lds_u32* p = (lds_u32*)(uint32_t)addr;
    uint32_t l[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) l[j] = p[j];
    uint32_t b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = __builtin_amdgcn_alignbyte(l[j + 1], l[j], sh);
    unsigned part = 0;
    if (PNORM == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) part = __builtin_amdgcn_sad_u8(arow[j], b[j], part);
    } else {
        unsigned bb = MSE_BIAS / 16, ab = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bb = __builtin_amdgcn_udot4(b[j], b[j], bb, false);
            ab = __builtin_amdgcn_udot4(arow[j], b[j], ab, false);
        }
        asm("s_nop 2\n\tv_mad_i32_i24 %0, %1, -2, %2" : "=v"(part) : "v"(ab), "v"(bb));
    }
    part = group8_sum(part);
    part += (unsigned)__builtin_amdgcn_update_dpp(0, (int)part, 0x128, 0xF, 0xF, false);     // row_ror 8: the other half of the row
    return part;
}

// minimum of one key per 16-lane row (every lane of a row holds its row's key) -> wave-uniform
__device__ __forceinline__ unsigned rows_min(unsigned key)
{
    key = min(key, (unsigned)__builtin_amdgcn_update_dpp((int)INF32, (int)key, 0x142, 0xA, 0xF, false));   // row_bcast15
    key = min(key, (unsigned)__builtin_amdgcn_update_dpp((int)INF32, (int)key, 0x143, 0xC, 0xF, false));   // row_bcast31
    return (unsigned)__builtin_amdgcn_readlane((int)key, 63);
}

// minimum of one key per 8-lane group (every lane of a group holds its group's key) -> wave-uniform: one DPP rotate
// inside each 16-lane row, rows 1,3 take lane 15 of the row below (row_bcast15), rows 2,3 take lane 31 (row_bcast31),
// lane 63 ends with the minimum
__device__ __forceinline__ unsigned groups_min(unsigned key)
{
    key = min(key, (unsigned)__builtin_amdgcn_update_dpp((int)INF32, (int)key, 0x128, 0xF, 0xF, false));   // row_ror 8
    key = min(key, (unsigned)__builtin_amdgcn_update_dpp((int)INF32, (int)key, 0x142, 0xA, 0xF, false));
    key = min(key, (unsigned)__builtin_amdgcn_update_dpp((int)INF32, (int)key, 0x143, 0xC, 0xF, false));
    return (unsigned)__builtin_amdgcn_readlane((int)key, 63);
}

// the part of [org - st, org + st] inside [0, limit]: holds every valid one of the three positions org - st, org,
// org + st of one axis (valid: 0 <= v <= limit); lo > hi when none of them is.  It may start before the first valid
// position (org - st < 0 <= org: the box starts at 0, not at org) -- the box only places the LDS window, which then
// covers a few rows more than it had to; two scalar instructions per bound instead of the fourteen of the exact range.
__device__ __forceinline__ void axis_box(int org, int st, int limit, int& lo, int& hi)
{
    lo = max(org - st, 0);
    hi = min(org + st, limit);
}

// What a wave fetches ahead for a block: its 2 anchor rows per lane and, for the diamond search (whose first
// window position depends on the block alone), the lane's share of that window.
template <int W> struct WalkPre {
    uint4 a0, a1;
    uint32_t w[W][4];       // the lane's share of the block's first window (W = segments per lane of the search's window)
    int wr0, wc0;           // that window's origin (wave-uniform)
    uint32_t mine;          // anchor dword (row lane / 4, dword lane % 4) for the all-lanes cost of one candidate (wave_eval_lds)
};

// first window of a diamond walk: the one PATTERN_MIN would stage around the clamped block origin
__device__ __forceinline__ void first_window(const WalkDev& d, int r0, int c0, int& wr0, int& wc0)
{
    wr0 = clamp_ref(r0, d.H - 17) - (WIN_ROWS - 16) / 2;
    wc0 = max(0, (clamp_ref(c0, d.W - 17) - (WIN_SPAN - 3) / 2) & ~3);          // never left of the frame, see window_load
}

// first window of a three-step / 2-D log walk whose patterns fit a window (FITS, see walk_block): the one EVALF would
// stage for the first round's box, the span `st` (first step / sw) around the block clipped to the frame
__device__ __forceinline__ void first_window_span(const WalkDev& d, int r0, int c0, int st, int& wr0, int& wc0)
{
    typedef Win<true> WS;
    int rmin, rmax, cmin, cmax;
    axis_box(r0, st, d.H - 16, rmin, rmax);
    axis_box(c0, st, d.W - 16, cmin, cmax);
    wr0 = rmin - ((WS::ROWS - 16 - (rmax - rmin)) >> 1);
    wc0 = max(0, (cmin - ((WS::SPAN - 3 - (cmax - cmin)) >> 1)) & ~3);
}

template <int PROC, bool FITS> constexpr int walk_pre()
{
    return PROC == GME_SEARCH_DIAMOND ? 1 : FITS && ((WALK_FITS_PREFETCH >> PROC) & 1) ? 2 : 0;
}

// PRE = 0: anchors only; 1: + the diamond's first window; 2: + the first window of a three-step / 2-D log walk (FITS)
template <int PRE, int W>
__device__ __forceinline__ void walk_prefetch(WalkPre<W>& f, const WalkDev& d, int pair, int brow, int bcol)
{
    const int lane = threadIdx.x & 63;
    const int r0 = brow * 16, c0 = bcol * 16;
    // anchors through a buffer resource as well: 32-bit offsets instead of 64-bit pointer arithmetic per lane
    const __amdgpu_buffer_rsrc_t ra = plane_rsrc(d.prev + (long long)pair * d.plane_stride, d.H * d.pitch);
    const int aoff = __mul24(r0 + (lane & 7), d.pitch) + c0;                    // block rows s and s + 8 (see walk_block)
    const u32x4_t t0 = __builtin_amdgcn_raw_buffer_load_b128(ra, aoff, 0, 0);
    const u32x4_t t1 = __builtin_amdgcn_raw_buffer_load_b128(ra, aoff + 8 * d.pitch, 0, 0);
    f.a0 = make_uint4(t0.x, t0.y, t0.z, t0.w); f.a1 = make_uint4(t1.x, t1.y, t1.z, t1.w);
    f.mine = __builtin_amdgcn_raw_buffer_load_b32(ra, __mul24(r0 + (lane >> 2), d.pitch) + c0 + 4 * (lane & 3), 0, 0);
    if constexpr (PRE == 1) {
        first_window(d, r0, c0, f.wr0, f.wc0);
        window_load<false>(f.w, d.cur + (long long)pair * d.plane_stride, d.pitch, d.H, f.wr0, f.wc0, lane);
    } else if constexpr (PRE == 2) {
        first_window_span(d, r0, c0, d.procedure == GME_SEARCH_THREESTEP ? d.st1 : d.sw, f.wr0, f.wc0);
        window_load<true>(f.w, d.cur + (long long)pair * d.plane_stride, d.pitch, d.H, f.wr0, f.wc0, lane);
    }
}

// one 16x16 block of one frame pair, walked by one wave; PROC = GME_SEARCH_THREESTEP / TWODLOG / DIAMOND.
// FITS (three-step, 2-D log): every round's box fits the 48 x 64 window (first step / sw <= 16), so there is no path
// that reads global memory directly, the first window is fetched ahead like the diamond's, and a round is "inside the
// staged window, else move it" -- about a fifth fewer scalar instructions per round (these kernels are bound by the
// scalar unit, profiles/r04_final_tdl720_pmc_summary.txt: 1.3 scalar instructions per vector instruction).
// `nxt`: the prefetch buffer of the wave's next block (brow_n, bcol_n), filled here if `more` -- after this block's
// fetched-ahead window has gone to the LDS, so that the two buffers' window registers are never live together.
template <int PNORM, int PROC, bool FITS, int NW>
__device__ __forceinline__ void walk_block(const WalkDev& d, const int pair, const int blk, const int brow, const int bcol,
                                           const int nblk, uint32_t* win, const WalkPre<NW>& pre,
                                           WalkPre<NW>& nxt, const bool more, const int brow_n, const int bcol_n)
{
    constexpr bool DIA = PROC == GME_SEARCH_DIAMOND;
    constexpr bool SBIG = !DIA && WALK_BIG;                      // window geometry of this search (Win<>)
    typedef Win<SBIG> WS;
    const long long gid = (long long)pair * nblk + blk;
    const int r0 = brow * 16, c0 = bcol * 16;
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 3, lrow = lane & 7;                 // group = candidate slot, lane = block rows lrow and lrow + 8
    const int H = d.H, W = d.W, pitch = d.pitch;
    const uint8_t* cur = d.cur + (long long)pair * d.plane_stride;

    uint32_t a[8];
    a[0] = pre.a0.x; a[1] = pre.a0.y; a[2] = pre.a0.z; a[3] = pre.a0.w;
    a[4] = pre.a1.x; a[5] = pre.a1.y; a[6] = pre.a1.z; a[7] = pre.a1.w;
    // MSE: a walk only compares costs of ONE block, so the block's own sum of a^2 -- the same in every candidate's
    // sum of (a - b)^2 = sum a^2 + sum b^2 - 2 sum a b -- is replaced by the constant 2^24 > 256 * 255^2: the order of the
    // costs, ties included, is unchanged, 8 v_dot4 per block (and one per all-lanes evaluation) are not issued.  Every
    // evaluation spreads the constant over its lanes (MSE_BIAS / lanes each); per-lane parts may wrap below zero, the
    // sum over the block cannot: 2^24 + sum b^2 - 2 sum a b >= 2^24 - sum a^2 > 0, and < 2^26, so keys still fit 32 bits.
    const unsigned aa = PNORM == 1 ? MSE_BIAS / 8 : 0u;

    // Candidates are served from the LDS window; when some fall outside it the window is moved (centred on their
    // bounding box) and, if the pattern is wider than the window (!FITS: sw > 16), the round reads global memory directly.
    int wr0 = -(1 << 20), wc0 = 0;                  // far away: nothing staged yet (the diamond and FITS walks set their first window themselves)
    bool round_lds = FITS;                          // the last EVALV round was served from the LDS window
#if defined(WALK_ABLATE) && WALK_ABLATE == 1      // timing experiments only (tools/build_variant.sh): anchors loaded, nothing else
    if (lane == 0) { int32_t* o = d.mf + gid * 2; o[0] = (int)(a[0] + a[7] + aa) >> 30; o[1] = 0; }
    return;
#endif
    // cost of this lane's group's candidate (RR, CC, OK: per-lane values, equal inside a group) -> COST in every lane of
    // the group (INF32 if !OK).  [RMIN, RMAX] x [CMIN, CMAX] is a wave-uniform box that holds the round's valid
    // candidates (RMIN > RMAX: none): when it fits, the round is served from the LDS window, which is moved (centred on
    // the box) if it does not cover it; a pattern wider than the window reads global memory directly (!FITS only).
    // These kernels are bound by the SCALAR unit -- wave-uniform booleans cost the compiler an
    // s_cmp + s_cselect_b64 each and an s_and_b64 per conjunction.  The tests are therefore sign tests of ORed
    // differences: "inside the staged window" is dr, rs - dr, dc, cs - dc all >= 0 (rs, cs: the slack of the box in a
    // window), "fits a window" is rs, cs - 3 >= 0 for a non-empty box; one s_or chain and one compare each.
    // FITS: a box always fits.  An empty box (a three-step origin that left the frame, bbme.py:332-336) may move the
    // window to a meaningless place: no lane reads it (nothing is valid), and the next round moves it again.
#define EVALV(RR, CC, OK, RMIN, RMAX, CMIN, CMAX, COST)                                               \
    do {                                                                                             \
        const int rh_ = (RMAX) - (RMIN), ch_ = (CMAX) - (CMIN);                                      \
        const int rs_ = (WS::ROWS - 16) - rh_, cs_ = WS::SPAN - ch_;                                 \
        const int dr_ = (RMIN) - wr0, dc_ = (CMIN) - wc0;                                            \
        bool lds_ok_ = true;                                                                         \
        if ((dr_ | (rs_ - dr_) | dc_ | (cs_ - dc_)) < 0) {                                           \
            if (FITS || (rh_ | ch_ | rs_ | (cs_ - 3)) >= 0) {                                        \
                wr0 = (RMIN) - (rs_ >> 1);                                                           \
                wc0 = max(0, ((CMIN) - ((cs_ - 3) >> 1)) & ~3);                                      \
                stage_walk_window<SBIG>(win, cur, pitch, H, wr0, wc0, lane);                         \
            } else lds_ok_ = false;                                                                  \
        }                                                                                            \
        if (FITS) {                                                                                  \
            COST = group_eval_lds<PNORM, SBIG>(a, aa, win, -wr0 * (4 * WS::PITCH) - wc0, RR, CC, OK, lrow);       \
        } else {                                                                                     \
            COST = lds_ok_ ? group_eval_lds<PNORM, SBIG>(a, aa, win, -wr0 * (4 * WS::PITCH) - wc0, RR, CC, OK, lrow)  \
                           : group_eval<PNORM>(a, aa, cur, pitch, RR, CC, OK, lrow);                 \
            round_lds = lds_ok_;                                                                     \
        }                                                                                            \
    } while (0)
    // the same for a round whose box the staged window is known to hold (FITS: the first round of a block)
#define EVALW(RR, CC, OK, COST)                                                                       \
    do { COST = group_eval_lds<PNORM, SBIG>(a, aa, win, -wr0 * (4 * WS::PITCH) - wc0, RR, CC, OK, lrow); } while (0)
    // the ninth candidate of a round (wave-uniform position inside the round's box): by all lanes from the window the
    // round's first EVALV has just made sure of, else by group 0 like the other eight
#define EVAL9(RR, CC, RMIN, RMAX, CMIN, CMAX, KEY)                                                    \
    do {                                                                                             \
        const bool ok9_ = (unsigned)(RR) <= (unsigned)(H - 16) && (unsigned)(CC) <= (unsigned)(W - 16);   \
        if (ok9_ && round_lds) {                                                                     \
            const unsigned c_ = wave_eval_lds<PNORM, WS::PITCH>(pre.mine, lds_address(win) - wr0 * (4 * WS::PITCH) - wc0, RR, CC, lane); \
            KEY = min(KEY, (c_ << 4) | 8u);                                                          \
        } else if (!FITS) {                                                                          \
            const bool ok_ = grp == 0 && ok9_;                                                       \
            unsigned c_;                                                                             \
            EVALV(RR, CC, ok_, RMIN, RMAX, CMIN, CMAX, c_);                                          \
            KEY = min(KEY, ok_ ? (c_ << 4) | 8u : INF32);                                            \
        }                                                                                            \
    } while (0)
    constexpr int PRE = walk_pre<PROC, FITS>();
    if constexpr (PRE != 0) {                         // fetched ahead by walk_prefetch
        wr0 = pre.wr0; wc0 = pre.wc0;
        window_store<SBIG>(win, pre.w, lane);
    }
    if (more) walk_prefetch<PRE>(nxt, d, pair, brow_n, bcol_n);     // in flight during this block's walk

    int out0 = 0, out1 = 0;
    bool overrun = false;
    const int cap = 2 * (H + W) + 64;

    if constexpr (DIA) {
        const int maxr = H - 16 - 1, maxc = W - 16 - 1;
        // (pr, pc): origin the pattern offsets are added to; (qr, qc): its clamped position = candidate 0 of the
        // pattern.  They differ only before the first round, when the block sits in the last block row / column
        // (r0 = H - 16 = maxr + 1); bbme.py:498-510 then spends one extra round moving onto the clamped position.
        int pr = r0, pc = c0;
        int qr = clamp_ref(r0, maxr), qc = clamp_ref(c0, maxc);
        bool raw = (qr != pr) | (qc != pc);
        unsigned centre_cost;
        // The window holds every pattern (|offset| <= 2, candidates clamped into the frame: within 2 of the clamped
        // centre) whose clamped centre lies in [cr_lo, cr_lo + 20] x [cc_lo, cc_lo + cc_span]; sbase is the
        // wave-uniform part of a candidate's LDS byte address (window base included).  Two unsigned compares per round instead of clamps.
        int cr_lo, cc_lo, cc_span, sbase;
#define WINDOW_RANGES()                                                                               \
    do {                                                                                             \
        cr_lo = wr0 + 2; cc_lo = wc0 == 0 ? 0 : wc0 + 2; cc_span = wc0 + WIN_SPAN - 2 - cc_lo;       \
        sbase = lds_address(win) - wr0 * (4 * WIN_PITCH) - wc0;                                      \
    } while (0)
#define RESTAGE_IF_OUTSIDE()                                                                          \
    do {                                                                                             \
        if (!((unsigned)(qr - cr_lo) <= (unsigned)(WIN_ROWS - 16 - 4) && (unsigned)(qc - cc_lo) <= (unsigned)cc_span)) { \
            const int rmin_ = max(qr - 2, 0), rmax_ = min(qr + 2, maxr);                             \
            const int cmin_ = max(qc - 2, 0), cmax_ = min(qc + 2, maxc);                             \
            wr0 = rmin_ - (WIN_ROWS - 16 - (rmax_ - rmin_)) / 2;                                     \
            wc0 = max(0, (cmin_ - (WIN_SPAN - 3 - (cmax_ - cmin_)) / 2) & ~3);                       \
            stage_walk_window(win, cur, pitch, H, wr0, wc0, lane);                                   \
            WINDOW_RANGES();                                                                         \
        }                                                                                            \
    } while (0)
        WINDOW_RANGES();                                // of the window fetched ahead (walk_prefetch), stored above
        // per-lane offsets of the large (groups 0..7) and small (groups 0..3) patterns, from nibble tables:
        // LDSP minus its centre (2,0),(1,1),(0,2),(-1,1),(-2,0),(-1,-1),(0,-2),(1,-1) as (row, col); the small
        // pattern's offsets are applied swapped (bbme.py:518-521): (0,1),(1,0),(0,-1),(-1,0)
        const int my_dr = (int)((0x32101234u >> (4 * grp)) & 15u) - 2, my_dc = (int)((0x10123432u >> (4 * grp)) & 15u) - 2;
        const int lrow52 = lrow * (4 * WIN_PITCH);
        // Winner of a pattern without leaving the vector unit: every lane of group k holds candidate k's
        // cost, key = cost << 3 | k (cost < 2^24), minimum over the 8 groups by one DPP rotate inside
        // each 16-lane row, two row broadcasts and one v_readlane.  Only a key whose cost is strictly below the centre's
        // moves the centre (bbme.py:507-510: first strict minimum, the centre is candidate 0); the
        // winner's offset comes out of the same nibble tables, so no per-candidate scalar position is needed.
#define PATTERN_MIN(n, OFF, SH, KMIN)                                                                  \
    do {                                                                                             \
        const unsigned c_ = group_eval_at<PNORM>(a, aa, OFF, SH, grp < (n));                        \
        KMIN = groups_min(grp < (n) ? (c_ << 3) | (unsigned)grp : INF32);                            \
    } while (0)
        {   // first centre: clamp(origin) (bbme.py:498-506), evaluated once, by all 64 lanes together: lane l takes
            // dword l % 4 of block row l / 4 (two window dwords, one v_alignbyte with a scalar shift, one v_sad_u8 or
            // two v_dot4), then 7 DPP adds leave the cost in lane 63 -- a third of the instructions of a
            // pattern round that would keep 7 of the 8 groups idle.  The prefetched window is built around it.
            centre_cost = wave_eval_lds<PNORM, WIN_PITCH>(pre.mine, sbase, qr, qc, lane);
        }
#if defined(WALK_ABLATE) && WALK_ABLATE == 2      // first window staged + centre evaluated
        if (lane == 0) { int32_t* o = d.mf + gid * 2; o[0] = (int)centre_cost >> 30; o[1] = 0; }
        return;
#endif
        int it = 0;
        bool again;
        do {
            RESTAGE_IF_OUTSIDE();
            unsigned kmin;
            {
                const int rrv = clamp_med3(pr + my_dr, maxr), ccv = clamp_med3(pc + my_dc, maxc);
                PATTERN_MIN(8, window_offset(sbase, rrv, ccv, lrow52), (uint32_t)ccv & 3u, kmin);
            }
            // A strictly better candidate moves the centre; its clamped position differs from the centre's (equal
            // positions have equal costs), so "the centre did not move" (bbme.py:511) is "nothing was better" --
            // except in the one round that starts from an unclamped origin.
            const bool better = (kmin >> 3) < centre_cost;
            if (better) {
                const unsigned k = kmin & 7u;
                qr = clamp_ref(pr + (int)((0x32101234u >> (4 * k)) & 15u) - 2, maxr);
                qc = clamp_ref(pc + (int)((0x10123432u >> (4 * k)) & 15u) - 2, maxc);
                centre_cost = kmin >> 3;
            }
            again = better | raw;
            raw = false;
            pr = qr; pc = qc;                                  // from here on the origin is a clamped position
        } while (again && ++it <= cap);
        overrun = again;
#if defined(WALK_ABLATE) && WALK_ABLATE == 3      // large-pattern rounds done, small pattern skipped
        if (lane == 0) { int32_t* o = d.mf + gid * 2; o[0] = pc - c0; o[1] = pr - r0; }
        return;
#endif
        {   // small pattern around the final centre
            RESTAGE_IF_OUTSIDE();
            int br = pr, bc = pc;
            // One candidate per 16-lane row, one block row per lane (row16_eval_at).  Offsets worked out here, once per
            // block, rather than kept in registers through the rounds (the volatile asm pins the computation to this
            // place: 64 VGPRs are what 8 waves per SIMD allow); the anchors of the large rounds are not needed any more,
            // so the lane's row overwrites a[0..3].
            int c4 = 4 * (lane >> 4);
            asm volatile("" : "+v"(c4));
            const int my_sr = (int)((0x0121u >> c4) & 15u) - 1, my_sc = (int)((0x1012u >> c4) & 15u) - 1;
            const int rrv = clamp_med3(pr + my_sr, maxr), ccv = clamp_med3(pc + my_sc, maxc);
            uint32_t arow[4];
            const uint32_t upper = 0u - (((uint32_t)lane >> 3) & 1u);          // all ones in lanes 8..15 of a row
#pragma unroll
            for (int j = 0; j < 4; ++j) arow[j] = (a[4 + j] & upper) | (a[j] & ~upper);     // v_bfi_b32 (a select between array elements became an indexed scratch access)
            const unsigned c_ = row16_eval_at<PNORM>(arow, window_offset(sbase, rrv, ccv, (lane & 15) * (4 * WIN_PITCH)), (uint32_t)ccv & 3u);
            const unsigned kmin = rows_min((c_ << 3) | (unsigned)(lane >> 4));
            if ((kmin >> 3) < centre_cost) {
                const unsigned k = kmin & 3u;
                br = clamp_ref(pr + (int)((0x0121u >> (4 * k)) & 15u) - 1, maxr);
                bc = clamp_ref(pc + (int)((0x1012u >> (4 * k)) & 15u) - 1, maxc);
            }
            out1 = br - r0; out0 = bc - c0;
        }
#undef PATTERN_MIN
#undef RESTAGE_IF_OUTSIDE
#undef WINDOW_RANGES
    } else if constexpr (PROC == GME_SEARCH_THREESTEP) {
        // bbme.py:182-341.  Candidate k = 0..8 of a step st around (org_r, org_c): column offset (k / 3 - 1) st in the outer
        // loop, row offset (k % 3 - 1) st in the inner one; out-of-frame candidates are skipped, the first strict
        // minimum wins.  Group g evaluates candidate g, candidate 8 is evaluated by all lanes together (EVAL9).
        // The centre of the SECOND step is the first step's winner (org = block + d1, bbme.py:260-261), a valid position
        // whose cost is known: its key (that cost, k = 4) joins the minimum as it is, group 4 evaluates candidate 8
        // instead and the step needs no ninth evaluation.  The same holds for the THIRD step when the first step stayed
        // where it was (the usual case on slow content).
        const int ur = grp % 3 - 1, uc = grp / 3 - 1;              // this group's offset in units of the step
        unsigned known = INF32;                                    // key of the step's centre (k = 4) when its cost is known
        int drow = 0, dcol = 0, trow = 0, tcol = 0, org_r = r0, org_c = c0;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int st = s == 0 ? d.st1 : s == 1 ? d.st2 : d.st3;
            int rmin, rmax, cmin, cmax;
            axis_box(org_r, st, H - 16, rmin, rmax);
            axis_box(org_c, st, W - 16, cmin, cmax);
            const bool have = s == 1 || (s == 2 && known != INF32);         // wave-uniform; s == 1: always
            const bool g4 = have && grp == 4;                               // group 4 takes candidate 8 then
            unsigned key;
            {
                const int rr = org_r + (g4 ? 1 : ur) * st, cc = org_c + (g4 ? 1 : uc) * st;
                const bool ok = (unsigned)rr <= (unsigned)(H - 16) && (unsigned)cc <= (unsigned)(W - 16);   // 0 <= v <= limit
                unsigned c;
                if (PRE != 0 && s == 0) EVALW(rr, cc, ok, c);          // the window fetched ahead is this round's
                else EVALV(rr, cc, ok, rmin, rmax, cmin, cmax, c);
                key = ok ? (c << 4) | (g4 ? 8u : (unsigned)grp) : INF32;
            }
            if (!have) EVAL9(org_r + st, org_c + st, rmin, rmax, cmin, cmax, key);        // candidate 8 (wave-uniform position)
            const unsigned kmin = min(groups_min(key), known);
            int kr = s == 0 ? drow : trow, kc = s == 0 ? dcol : tcol;       // nothing valid: the stale offsets stay (bbme.py:332-336)
            if (kmin != INF32) {
                const int k = (int)(kmin & 15u);
                kr = (k % 3 - 1) * st; kc = (k / 3 - 1) * st;
            }
            if (s == 0) {
                drow = kr; dcol = kc; org_r = r0 + drow; org_c = c0 + dcol;
                known = (kmin & ~15u) | 4u;                         // the block's own position is always valid: kmin != INF32
            } else {
                // the third step's origin is org + (d1 + d2), the accumulated offset (bbme.py:332-336): the second step's
                // winner org + d2 -- the one position whose cost is at hand -- only when the first step did not move
                if (s == 1) known = (drow | dcol) == 0 ? (kmin & ~15u) | 4u : INF32;
                trow = kr; tcol = kc; drow += trow; dcol += tcol; org_r += drow; org_c += dcol;
            }
        }
        out0 = dcol; out1 = drow;
    } else {   // 2-D log, bbme.py:344-433
        // step > 2: centre, (+s, 0), (-s, 0), (0, +s), (0, -s) as (row, col) offsets, groups 0..4; step == 2: the ring
        // (k / 3 - 1, k % 3 - 1) * 2 for k = 0..8.  The ring's centre (k = 4) is where the last cross round ended, its
        // cost known from that round: group 4 takes candidate 8 and the centre's key joins the minimum as it is.  Only a
        // search that starts at step 2 (sw = 2: no cross round) evaluates the ninth candidate by all lanes (EVAL9).
        const int xr = grp == 1 ? 1 : grp == 2 ? -1 : 0;
        const int xc = grp == 3 ? 1 : grp == 4 ? -1 : 0;
        int br = 0, bc = 0, pr = r0, pc = c0, step = d.sw, it = 0;
        unsigned known = INF32;
        const unsigned rlim = (unsigned)(H - 16), clim = (unsigned)(W - 16);      // 0 <= v <= limit as one unsigned compare
        // The cross rounds (step > 2) and the one ring round that ends the walk (step == 2: bbme.py halves the step after
        // it whatever it found) are two pieces of code: no per-round selects between the two patterns.
        while (step > 2) {
            int rmin, rmax, cmin, cmax;
            axis_box(pr, step, H - 16, rmin, rmax);
            axis_box(pc, step, W - 16, cmin, cmax);
            const int rr = pr + xr * step, cc = pc + xc * step;
            const bool ok = grp < 5 && (unsigned)rr <= rlim && (unsigned)cc <= clim;
            unsigned c;
            EVALV(rr, cc, ok, rmin, rmax, cmin, cmax, c);
            const unsigned kmin = groups_min(ok ? (c << 4) | (unsigned)grp : INF32);
            known = kmin;                                           // the centre is always valid, so never INF32
            {
                const int k = (int)(kmin & 15u);
                br = pr + (k == 1 ? step : k == 2 ? -step : 0); bc = pc + (k == 3 ? step : k == 4 ? -step : 0);
            }
            if (br == pr && bc == pc) step /= 2;
            pr = br; pc = bc;
            if (++it > cap) { overrun = true; break; }
        }
        if (step == 2 && !overrun) {
            const bool have = known != INF32;                       // wave-uniform
            known = have ? (known & ~15u) | 4u : INF32;
            const bool g4 = have && grp == 4;
            const int gr = g4 ? 1 : grp / 3 - 1, gc = g4 ? 1 : grp % 3 - 1;
            int rmin, rmax, cmin, cmax;
            axis_box(pr, 2, H - 16, rmin, rmax);
            axis_box(pc, 2, W - 16, cmin, cmax);
            unsigned key;
            {
                const int rr = pr + gr * 2, cc = pc + gc * 2;
                const bool ok = (unsigned)rr <= rlim && (unsigned)cc <= clim;
                unsigned c;
                EVALV(rr, cc, ok, rmin, rmax, cmin, cmax, c);
                key = ok ? (c << 4) | (g4 ? 8u : (unsigned)grp) : INF32;
            }
            if (!have) EVAL9(pr + 2, pc + 2, rmin, rmax, cmin, cmax, key);
            const unsigned kmin = min(groups_min(key), known);
            if (kmin != INF32) {
                const int k = (int)(kmin & 15u);
                br = pr + (k / 3 - 1) * 2; bc = pc + (k % 3 - 1) * 2;
            }
            if (++it > cap) overrun = true;
        }
        out1 = br - r0; out0 = bc - c0;
    }
#undef EVAL9
#undef EVALW
#undef EVALV
    if (lane == 0) {
        if (overrun) atomicExch(d.status, 1);
        int32_t* o = d.mf + gid * 2;
        o[0] = out0; o[1] = out1;
    }
}

// Workgroup = 4 waves, each wave walks d.bpw blocks one after the other (blocks base, base + 4, ...): the
// dispatcher starts ~2 waves per clock chip-wide, which at one short walk per wave was a fifth of the kernel's time.
// A wave keeps its block's (row, column) in the block grid and steps it by 4 columns: the two integer divisions per
// block that `blk / nbc`, `blk % nbc` cost were 45 of the ~260 scalar instructions of a three-step block.
__device__ __forceinline__ void next_block(int& brow, int& bcol, int nbc)
{
    bcol += 4;
    while (bcol >= nbc) { bcol -= nbc; ++brow; }                // nbc >= 4: at most once
}

template <int PNORM, int PROC, bool FITS, int ALLOC>
__device__ __forceinline__ void walk16_workgroup(const WalkDev& d, uint32_t (&win_all)[4][ALLOC])
{
    constexpr bool DIA = PROC == GME_SEARCH_DIAMOND;
    constexpr int PRE = walk_pre<PROC, FITS>(), NW = PRE == 1 ? Win<false>::SEGS_LANE : PRE == 2 ? Win<true>::SEGS_LANE : 1;
    static_assert(!FITS || (!DIA && WALK_BIG), "FITS is a property of the 48 x 64 window");
    const int wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nblk = d.nbr * d.nbc;
    // XCD-aware: workgroups are dealt round-robin over the 8 XCDs, so workgroup b serves pair
    // (b/8/WPP)*8 + b%8 -- all blocks of a frame pair walk through one XCD's L2
    const int per = 4 * d.bpw, wpp = (nblk + per - 1) / per;
    const int pair = ((int)blockIdx.x / 8 / wpp) * 8 + ((int)blockIdx.x & 7);
    const int base = (((int)blockIdx.x >> 3) % wpp) * per + wave_in_wg;
    if (pair >= d.pairs) return;                               // wave-uniform
    if (base >= nblk) return;
    // two prefetch buffers used in turn (the loop is unrolled by two so that no registers are copied around)
    WalkPre<NW> pa, pb;
    int ra = base / d.nbc, ca = base - ra * d.nbc, rb, cb;     // block-grid position of pa's block, pb's block
    walk_prefetch<PRE>(pa, d, pair, ra, ca);
    for (int i = 0; i < d.bpw; i += 2) {
        const int blk = base + 4 * i;                               // valid: checked before it was fetched
        const bool more1 = i + 1 < d.bpw && blk + 4 < nblk;
        rb = ra; cb = ca; next_block(rb, cb, d.nbc);
        walk_block<PNORM, PROC, FITS>(d, pair, blk, ra, ca, nblk, win_all[wave_in_wg], pa, pb, more1, rb, cb);
        if (!more1) break;
        const bool more2 = i + 2 < d.bpw && blk + 8 < nblk;
        ra = rb; ca = cb; next_block(ra, ca, d.nbc);
        walk_block<PNORM, PROC, FITS>(d, pair, blk + 4, rb, cb, nblk, win_all[wave_in_wg], pb, pa, more2, ra, ca);
        if (!more2) break;
    }
}

// The diamond instance is held to 64 VGPRs (8 waves per SIMD cover its LDS -> dot4 -> DPP -> v_readlane -> scalar chain);
// three-step and 2-D log fit 8 waves anyway, and without the cap their scalar state needs no spill.
template <int PNORM>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) k_walk16(WalkDev d)
{
    __shared__ uint32_t win_all[4][WIN_ALLOC];
    walk16_workgroup<PNORM, GME_SEARCH_DIAMOND, false, WIN_ALLOC>(d, win_all);
}

// waves per SIMD the register allocation aims at: 8, or 6 for an instance that holds a prefetched window (80 VGPRs)
#ifndef WALK16S_WAVES_PRE
#define WALK16S_WAVES_PRE 6
#endif
// FITS: first step (three-step) / sw (2-D log) <= 16 -- every search the pipeline and the benches run; wider searches
// take the instance with the global-memory path
template <int PNORM, int PROC, bool FITS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(!(FITS && WALK_BIG) ? 1 : walk_pre<PROC, true>() == 2 ? WALK16S_WAVES_PRE : 8, walk_pre<PROC, FITS && WALK_BIG>() == 2 ? WALK16S_WAVES_PRE : 8))) k_walk16s(WalkDev d)
{
    __shared__ uint32_t win_all[4][Win<WALK_BIG>::ALLOC];
    walk16_workgroup<PNORM, PROC, FITS && WALK_BIG, Win<WALK_BIG>::ALLOC>(d, win_all);
}

// ---------------------------------------------------------------------------
// dense 2x2 diamond search, one lane per block
// ---------------------------------------------------------------------------
template <int PNORM>
__device__ __forceinline__ unsigned cost2x2(uint32_t a, unsigned aa, const uint8_t* cur, int pitch, int rr, int cc)
{
    const uint8_t* p = cur + (long long)rr * pitch + cc;
    const uint32_t b = (uint32_t)(*(const u16_u*)p) | ((uint32_t)(*(const u16_u*)(p + pitch)) << 16);
    if (PNORM == 0) return __builtin_amdgcn_sad_u8(a, b, 0u);
    return aa + __builtin_amdgcn_udot4(b, b, 0u, false) - 2u * __builtin_amdgcn_udot4(a, b, 0u, false);
}

template <int PNORM>
__global__ void __launch_bounds__(256) k_dense2(WalkDev d)
{
    const int nblk = d.nbr * d.nbc;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)nblk * d.pairs) return;
    const int pair = (int)(gid / nblk), blk = (int)(gid % nblk);
    const int r0 = (blk / d.nbc) * 2, c0 = (blk % d.nbc) * 2;
    const int pitch = d.pitch;
    const uint8_t* cur = d.cur + (long long)pair * d.plane_stride;
    const uint8_t* ap = d.prev + (long long)pair * d.plane_stride + (long long)r0 * pitch + c0;
    const uint32_t a = (uint32_t)(*(const uint16_t*)ap) | ((uint32_t)(*(const uint16_t*)(ap + pitch)) << 16);
    const unsigned aa = PNORM ? __builtin_amdgcn_udot4(a, a, 0u, false) : 0u;
    const int maxr = d.H - 2 - 1, maxc = d.W - 2 - 1;
    const int ldr[8] = { 2, 1, 0, -1, -2, -1, 0, 1 }, ldc[8] = { 0, 1, 2, 1, 0, -1, -2, -1 };
    int pr = r0, pc = c0;
    unsigned centre = cost2x2<PNORM>(a, aa, cur, pitch, clamp_ref(pr, maxr), clamp_ref(pc, maxc));
    const int cap = 2 * (d.H + d.W) + 64;
    bool overrun = false;
    for (int it = 0;; ++it) {
        unsigned best = centre;
        int br = clamp_ref(pr, maxr), bc = clamp_ref(pc, maxc);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int rr = clamp_ref(pr + ldr[k], maxr), cc = clamp_ref(pc + ldc[k], maxc);
            const unsigned c = cost2x2<PNORM>(a, aa, cur, pitch, rr, cc);
            if (c < best) { best = c; br = rr; bc = cc; }
        }
        const bool done = (br == pr && bc == pc);
        pr = br; pc = bc; centre = best;
        if (done) break;
        if (it > cap) { overrun = true; break; }
    }
    const int sdr[4] = { 0, 1, 0, -1 }, sdc[4] = { 1, 0, -1, 0 };
    unsigned best = centre;
    int br = pr, bc = pc;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int rr = clamp_ref(pr + sdr[k], maxr), cc = clamp_ref(pc + sdc[k], maxc);
        const unsigned c = cost2x2<PNORM>(a, aa, cur, pitch, rr, cc);
        if (c < best) { best = c; br = rr; bc = cc; }
    }
    if (overrun) atomicExch(d.status, 1);
    int32_t* o = d.mf + gid * 2;
    o[0] = bc - c0; o[1] = br - r0;
}

}  // namespace

int launch_bbme_walk_fast(gme_ctx* ctx, const BbmeJob& job, bool* handled)
{
    *handled = false;
    if (job.procedure == GME_SEARCH_EXHAUSTIVE) return GME_OK;
    const int nbr = job.H / job.bs, nbc = job.W / job.bs;
    if (nbr == 0 || nbc == 0 || job.pairs == 0) return GME_OK;
    WalkDev d;
    d.prev = job.prev; d.cur = job.cur; d.plane_stride = job.plane_stride; d.pairs = job.pairs;
    d.H = job.H; d.W = job.W; d.pitch = job.pitch; d.sw = job.sw; d.procedure = job.procedure;
    d.nbr = nbr; d.nbc = nbc; d.mf = job.mf; d.status = ctx->status;
    const int span = 2 * job.sw + job.bs;
    d.st1 = (int)(span / 3.0); d.st2 = (int)(span / 5.0); d.st3 = (int)(span / 10.0);
    const long long total = (long long)nbr * nbc * job.pairs;
    if (job.bs == 16) {
        static const int bpw_env = getenv("GME_WALK_BPW") ? atoi(getenv("GME_WALK_BPW")) : 0;
        d.bpw = bpw_env >= 1 && bpw_env <= 64 ? bpw_env : 8;
        const long long wpp = ((long long)nbr * nbc + 4 * d.bpw - 1) / (4 * d.bpw);
        const long long groups = (long long)((job.pairs + 7) / 8) * 8 * wpp;
        GME_REQUIRE(groups < (1ll << 31), GME_ERR_ARG, "too many workgroups in one launch");
        const unsigned grid = (unsigned)groups;
        // one instance per search and norm: the diamond alone needs half the scalar state of a body that holds all three
        if (job.procedure == GME_SEARCH_DIAMOND) {
            plan_note(ctx, 0, "k_walk16<%d> (diamond) grid %u blocks/wave %d", job.pnorm, (unsigned)grid, d.bpw);
            if (job.pnorm == 0) hipLaunchKernelGGL(k_walk16<0>, dim3(grid), dim3(256), 0, ctx->stream, d);
            else hipLaunchKernelGGL(k_walk16<1>, dim3(grid), dim3(256), 0, ctx->stream, d);
        } else {
            const bool tss = job.procedure == GME_SEARCH_THREESTEP;
            // the widest round's span (2 x first step / 2 x sw) must fit the window's 32 rows of slack (Win<true>)
            const bool fits = WALK_BIG && (tss ? d.st1 : d.sw) <= (Win<true>::ROWS - 16) / 2;
            plan_note(ctx, 0, "k_walk16s<%d,%d,%s> (%s) grid %u blocks/wave %d", job.pnorm, job.procedure, fits ? "true" : "false", tss ? "three-step" : "2-D log", (unsigned)grid, d.bpw);
#define LAUNCH_WALK16S(P, S)                                                                                     \
    do {                                                                                                         \
        if (fits) hipLaunchKernelGGL((k_walk16s<P, S, true>), dim3(grid), dim3(256), 0, ctx->stream, d);         \
        else hipLaunchKernelGGL((k_walk16s<P, S, false>), dim3(grid), dim3(256), 0, ctx->stream, d);             \
    } while (0)
            if (tss && job.pnorm == 0) LAUNCH_WALK16S(0, GME_SEARCH_THREESTEP);
            else if (tss) LAUNCH_WALK16S(1, GME_SEARCH_THREESTEP);
            else if (job.pnorm == 0) LAUNCH_WALK16S(0, GME_SEARCH_TWODLOG);
            else LAUNCH_WALK16S(1, GME_SEARCH_TWODLOG);
#undef LAUNCH_WALK16S
        }
    } else if (job.bs == 2 && job.procedure == GME_SEARCH_DIAMOND) {
        const unsigned grid = (unsigned)((total + 255) / 256);
        plan_note(ctx, 0, "k_dense2<%d> grid %u", job.pnorm, (unsigned)grid);
        if (job.pnorm == 0) hipLaunchKernelGGL(k_dense2<0>, dim3(grid), dim3(256), 0, ctx->stream, d);
        else hipLaunchKernelGGL(k_dense2<1>, dim3(grid), dim3(256), 0, ctx->stream, d);
    } else {
        return GME_OK;
    }
    GME_HIP_TRY(hipGetLastError());
    *handled = true;
    return GME_OK;
}
