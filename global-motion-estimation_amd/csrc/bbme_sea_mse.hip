// Exhaustive MSE search, bs = 16, with exact successive elimination -- same answer as
// k_exh_dot16 (bbme_fast.hip) / bbme.py:105-179 with pnorm 1, a fraction of the work.
//
// Bound.  For each 8x8 quadrant q, (sum(A_q) - sum(B_q))^2 <= 64 * SSD_q (Cauchy-Schwarz on the 64
// pixel differences), hence  LBx = sum_q dS_q^2 <= 64 * SSD.  With a real candidate's SSD as UB,
// a candidate with LBx > 64 * UB has SSD > UB >= min and can neither win nor tie.  Bounds travel
// as floor(LBx / 32) (25 bits: LBx < 2^30) against floor(64 * UB / 32) = 2 UB -- half an SSD unit of slack; comparing
// floors only ever keeps more candidates, never fewer.  (Round 2 compared floor(LBx / 2^14) with UB >> 8: at the
// UB of a good match, a few hundred, that let bounds up to 1.5 x the exact limit through.)  List entries keep 19 bits,
// floor(LBx / 2^11), for the re-check against a tightened UB (UB >> 5).
//
// Same phases as k_exh_sea16 (bbme_sea.hip); what differs:
//   B  dS_q by v_pk_sub_i16 on the packed quadrant sums, squares summed by v_dot2_i32_i16;
//   C  UB from two cooperative SSDs (sum a^2 + sum b^2 - 2 sum ab with v_dot4_u32_u8 per lane);
//   E  a listed patch costs 16*R*16 v_dot4_u32_u8 (three v_alignbyte copies per window dword) for
//      sum(A.B); sum(B^2) comes from the per-frame 16x16 box table of squares (k_sqbox16_*),
//      sum(A^2) from phase A; 37-bit keys (ssd << 13 | scan index) merged with 64-bit LDS atomicMin.
#include "bbme_sea_common.h"

namespace {

using namespace sea;

typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned long long u64min_(unsigned long long a, unsigned long long b) { return a < b ? a : b; }


// per 16-bit half: a * b + c  (the compiler splits `a * 2 - c` into a shift and a subtraction)
__device__ __forceinline__ uint32_t pk_mad_i16(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c));
    return d;
}

// B: as lower_bounds() of bbme_sea.hip, with the squared bound: key = floor(sum_q dS_q^2 / 32) << 7 | local index.
template <int R, bool GUARD>
__device__ __forceinline__ void lower_bounds_mse(const uint64_t* sp0, int XQ, int prow, int q, uint32_t a01, uint32_t a23,
                                                 int lo_r, int hi_r, int lo_c, int hi_c, uint32_t (&pkey)[R])
{
#pragma unroll
    for (int k = 0; k < R; ++k) pkey[k] = 0xFFFFFFFFu;
    const uint64_t* top = sp0;
    const uint64_t* bot = sp0 + 8 * XQ;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int ri = prow * R + i;
        if (!GUARD || (ri >= lo_r && ri <= hi_r)) {
            uint64_t t[R + 2], b[R + 2];
#pragma unroll
            for (int k = 0; k < R + 2; ++k) { t[k] = top[k]; b[k] = bot[k]; }
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int ci0 = q * 4 * R + 4 * k;
                if (GUARD && (ci0 > hi_c || ci0 + 3 < lo_c)) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (GUARD && (ci0 + e < lo_c || ci0 + e > hi_c)) continue;
                    const uint32_t sel = (e & 1) ? 0x07060302u : 0x05040100u;
                    const uint32_t tp = __builtin_amdgcn_perm((uint32_t)(t[k + 2] >> (32 * (e >> 1))), (uint32_t)(t[k] >> (32 * (e >> 1))), sel);
                    const uint32_t bt = __builtin_amdgcn_perm((uint32_t)(b[k + 2] >> (32 * (e >> 1))), (uint32_t)(b[k] >> (32 * (e >> 1))), sel);
                    // twice the differences (one v_pk_mad_i16 each, as many instructions as the subtraction): the squares then
                    // sum to 4 LBx < 2^32, whose bits 7.. ARE floor(LBx / 32) -- one v_and_or builds the key where
                    // (lbx >> 5) << 7 needed a shift and a shift-add (round 4: 9 -> 8 instructions per candidate)
                    const s16x2 dt = __builtin_bit_cast(s16x2, pk_mad_i16(tp, 0x00020002u, a01));    // a01, a23: MINUS twice the anchor's sums; |d| <= 32640
                    const s16x2 db = __builtin_bit_cast(s16x2, pk_mad_i16(bt, 0x00020002u, a23));
                    // clamp = true only to get the VOP3P form (the VOP2 form v_dot2c needs its accumulator moved into place: a
                    // v_mov per candidate); saturation at 2^31 - 1 -- reachable only where 4 LBx >= 2^31, far above any upper
                    // bound -- makes a bound smaller, never larger, so it stays a lower bound
                    const uint32_t lbx4 = (uint32_t)__builtin_amdgcn_sdot2(dt, dt, __builtin_amdgcn_sdot2(db, db, 0, true), true);
                    pkey[k] = min(pkey[k], (lbx4 & ~127u) | (uint32_t)((4 * k + e) * R + i));          // local < 4R*R <= 100 < 128
                }
            }
        }
        top += XQ;
        bot += XQ;
    }
}

#ifndef SEA_MSE_TWO_LEVEL_FROM
#define SEA_MSE_TWO_LEVEL_FROM 4      // window classes R >= this use the two-level bound (R = 5, sw 32: +6 %; R = 3, sw 16: +-0, real frames -3 %)
#endif
// Level 1 of the two-level bound (R >= SEA_MSE_TWO_LEVEL_FROM): the MAE kernel's quadrant bound L1 = sum_q |dS_q| (5.7 instructions per
// candidate against 8 for the squared form).  SSD >= SAD^2 / 256 >= L1^2 / 256 (Cauchy-Schwarz over the 256 pixels, then the
// triangle inequality per quadrant), so a patch whose smallest L1 has L1^2 > 256 UB holds no winner; the squared form --
// never weaker -- is applied as level 2 to the patches that pass, one lane per patch (bounds2_patch).
template <int R, bool GUARD>
__device__ __forceinline__ void lower_bounds_l1(const uint64_t* sp0, int XQ, int prow, int q, uint32_t a01, uint32_t a23,
                                                int lo_r, int hi_r, int lo_c, int hi_c, uint32_t (&pkey)[R])
{
#pragma unroll
    for (int k = 0; k < R; ++k) pkey[k] = 0xFFFFFFFFu;
    const uint64_t* top = sp0;
    const uint64_t* bot = sp0 + 8 * XQ;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int ri = prow * R + i;
        if (!GUARD || (ri >= lo_r && ri <= hi_r)) {
            uint64_t t[R + 2], b[R + 2];
#pragma unroll
            for (int k = 0; k < R + 2; ++k) { t[k] = top[k]; b[k] = bot[k]; }
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int ci0 = q * 4 * R + 4 * k;
                if (GUARD && (ci0 > hi_c || ci0 + 3 < lo_c)) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (GUARD && (ci0 + e < lo_c || ci0 + e > hi_c)) continue;
                    const uint32_t sel = (e & 1) ? 0x07060302u : 0x05040100u;
                    const uint32_t tp = __builtin_amdgcn_perm((uint32_t)(t[k + 2] >> (32 * (e >> 1))), (uint32_t)(t[k] >> (32 * (e >> 1))), sel);
                    const uint32_t bt = __builtin_amdgcn_perm((uint32_t)(b[k + 2] >> (32 * (e >> 1))), (uint32_t)(b[k] >> (32 * (e >> 1))), sel);
                    const uint32_t lb = __builtin_amdgcn_sad_u16(tp, a01, __builtin_amdgcn_sad_u16(bt, a23, 0u));
                    pkey[k] = min(pkey[k], (lb << 13) + (uint32_t)((4 * k + e) * R + i));
                }
            }
        }
        top += XQ;
        bot += XQ;
    }
}

// Level 2: floor(LBx / 32) of the best of a patch's 4 x R candidates (the squared quadrant bound of lower_bounds_mse), by ONE
// lane for a patch of any block of the tile.  No validity guards: a candidate outside the frame can only make the patch's
// bound smaller (the patch is then scored for nothing, never dropped wrongly); phase E builds keys from valid candidates only.
template <int R>
__device__ __forceinline__ uint32_t bounds2_patch(const uint64_t* sp, int XQ, int k, uint32_t a01m, uint32_t a23m)
{
    uint32_t best = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const uint64_t t0 = sp[i * XQ + k], t2 = sp[i * XQ + k + 2], b0 = sp[(i + 8) * XQ + k], b2 = sp[(i + 8) * XQ + k + 2];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t sel = (e & 1) ? 0x07060302u : 0x05040100u;
            const uint32_t tp = __builtin_amdgcn_perm((uint32_t)(t2 >> (32 * (e >> 1))), (uint32_t)(t0 >> (32 * (e >> 1))), sel);
            const uint32_t bt = __builtin_amdgcn_perm((uint32_t)(b2 >> (32 * (e >> 1))), (uint32_t)(b0 >> (32 * (e >> 1))), sel);
            const s16x2 dt = __builtin_bit_cast(s16x2, pk_mad_i16(tp, 0x00020002u, a01m));
            const s16x2 db = __builtin_bit_cast(s16x2, pk_mad_i16(bt, 0x00020002u, a23m));
            best = min(best, (uint32_t)__builtin_amdgcn_sdot2(dt, dt, __builtin_amdgcn_sdot2(db, db, 0, true), true));       // clamp: see lower_bounds_mse
        }
    }
    return best >> 7;                                      // 4 LBx >> 7 = floor(LBx / 32)
}

// Phases A' .. F of one tile (entry conditions as tile_phases() of bbme_sea.hip, plus a2s[] filled).
template <int R, int LPPT>
__device__ __forceinline__ bool tile_phases_mse(const SeaDev& d, uint32_t* lds, const Layout& L, int pair, int trow, int bcol0,
                                                uint32_t mine, uint32_t a01, uint32_t a23, uint32_t mine2, int tid, int tile_id)
{
    constexpr bool TWO = R >= SEA_MSE_TWO_LEVEL_FROM;     // two-level bound: L1 form for every candidate, squared form for the listed patches
    const int T = blockDim.x;
    const int NC = 2 * d.sw + 16, XQ = d.xq;
    uint32_t* win = lds + L.win;
    uint32_t* anchor = lds + L.anchor;
    unsigned long long* best = (unsigned long long*)(lds + L.best);     // [NB] ssd << 13 | scan index
    uint32_t* count = lds + L.count;
    uint32_t* a2s = lds + L.a2;                                         // [NB] sum of squares of each anchor
    uint64_t* s8 = (uint64_t*)(lds + L.s8);
    uint32_t* work = lds + L.work;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const WaveBlock wb = wave_block(d, trow, bcol0, wave);
    const int brow = wb.brow, bcol = wb.bcol;
    const bool wave_ok = wb.ok;
    const int r0 = brow * 16, c0 = bcol * 16;
    const int prow = lane >> 2, q = lane & 3;

    // ---- A'
    box_sums8<R>(d, win, s8, tid);
    __syncthreads();

    // ---- B
    const int lo_r = max(0, d.sw - r0), hi_r = min(NC - 1, d.H - 16 - r0 + d.sw);
    const bool rows_inside = NC == 16 * R && lo_r == 0 && hi_r == NC - 1;   // and no padding candidates
    if (wave_ok) {
        const int lo_c = max(0, d.sw - c0), hi_c = min(NC - 1, d.W - 16 - c0 + d.sw);
        uint32_t pkey[R];
        const uint64_t* sp0 = s8 + (16 * wb.wr + prow * R) * XQ + wb.wc * 4 + q * R;
        uint32_t patch_lb[R], lb_key = 0xFFFFFFFFu;        // patch_lb: the patch's smallest bound (TWO: L1; else floor(LBx / 32), 25 bits)
        if constexpr (TWO) {
            // plain quadrant sums back from their "minus twice" form (prep): halves are multiples of 2 <= 32640 in magnitude
            const uint32_t p01 = __builtin_bit_cast(uint32_t, (s16x2)(-__builtin_bit_cast(s16x2, a01))) >> 1 & 0x7FFF7FFFu;
            const uint32_t p23 = __builtin_bit_cast(uint32_t, (s16x2)(-__builtin_bit_cast(s16x2, a23))) >> 1 & 0x7FFF7FFFu;
            if (rows_inside && lo_c == 0 && hi_c == NC - 1)                                // wave-uniform
                lower_bounds_l1<R, false>(sp0, XQ, prow, q, p01, p23, lo_r, hi_r, lo_c, hi_c, pkey);
            else
                lower_bounds_l1<R, true>(sp0, XQ, prow, q, p01, p23, lo_r, hi_r, lo_c, hi_c, pkey);
            if (lane == 0) { lds[L.own + 2 * wave] = a01; lds[L.own + 2 * wave + 1] = a23; }      // level 2 serves any block of the tile
#pragma unroll
            for (int k = 0; k < R; ++k) {
                patch_lb[k] = pkey[k] == 0xFFFFFFFFu ? 0xFFFFFFFFu : pkey[k] >> 13;
                lb_key = min(lb_key, pkey[k] == 0xFFFFFFFFu ? 0xFFFFFFFFu : ((pkey[k] >> 13) << 7) | (pkey[k] & 127u));   // bound << 7 | local, as below
            }
        } else {
            if (rows_inside && lo_c == 0 && hi_c == NC - 1)                                // wave-uniform
                lower_bounds_mse<R, false>(sp0, XQ, prow, q, a01, a23, lo_r, hi_r, lo_c, hi_c, pkey);
            else
                lower_bounds_mse<R, true>(sp0, XQ, prow, q, a01, a23, lo_r, hi_r, lo_c, hi_c, pkey);
#pragma unroll
            for (int k = 0; k < R; ++k) {
                patch_lb[k] = pkey[k] == 0xFFFFFFFFu ? 0xFFFFFFFFu : pkey[k] >> 7;      // floor(LBx / 32), 25 bits
                lb_key = min(lb_key, pkey[k]);
            }
        }
        // ---- C
        const uint32_t lb_min = wave_min_u32(lb_key);      // the zero vector is always valid: never the sentinel
        unsigned long long ub_key = ~0ull;
        {
            // the key has no room for the lane number: the first lane that holds the minimum names the candidate
            const int bl = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(lb_key == lb_min)) - 1), loc = lb_min & 127;
            const int ce = loc / R, li = loc - ce * R;
            const int idx1 = ((bl & 3) * 4 * R + ce) * NC + (bl >> 2) * R + li;
            const int idx0 = d.sw * NC + d.sw;
            const int arow = lane >> 2, aj = lane & 3;
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                const int idx = which ? idx1 : idx0;
                const int ci = idx / NC, ri = idx - ci * NC;
                const int byte = wb.wc * 16 + ci + 4 * aj;
                const uint32_t* p = win + (16 * wb.wr + ri + arow) * d.pitch_dw + (byte >> 2);
                const uint32_t v = __builtin_amdgcn_alignbyte(p[1], p[0], (uint32_t)byte & 3u);
                const uint32_t part = mine2 + __builtin_amdgcn_udot4(v, v, 0u, false) - 2u * __builtin_amdgcn_udot4(v, mine, 0u, false);
                const uint32_t ssd = wave_sum_u32(part);
                ub_key = u64min_(ub_key, ((unsigned long long)ssd << 13) | (unsigned)idx);
            }
            // third probe, as in the MAE kernel (bbme_sea.hip): the vector this wave's block of the previous tile ended with
            const int idxp = (int)(lds[L.prev + wave] & 0x1FFFu);
            const int cip = idxp / NC, rip = idxp - cip * NC;
            if (idxp != idx0 && idxp != idx1 && cip >= lo_c && cip <= hi_c && rip >= lo_r && rip <= hi_r) {
                const int byte = wb.wc * 16 + cip + 4 * aj;
                const uint32_t* p = win + (16 * wb.wr + rip + arow) * d.pitch_dw + (byte >> 2);
                const uint32_t v = __builtin_amdgcn_alignbyte(p[1], p[0], (uint32_t)byte & 3u);
                const uint32_t part = mine2 + __builtin_amdgcn_udot4(v, v, 0u, false) - 2u * __builtin_amdgcn_udot4(v, mine, 0u, false);
                ub_key = u64min_(ub_key, ((unsigned long long)wave_sum_u32(part) << 13) | (unsigned)idxp);
            }
        }
        if (lane == 0) best[wave] = ub_key;
        // ---- D
        const uint32_t ub25 = (uint32_t)(ub_key >> 12);                // 2 ssd = floor(64 * ssd / 32); ssd < 2^24
        // UB == 0 (an exact match is known): only another exact match earlier in scan order can replace it
        // (bbme.py:171 keeps the first minimum), so a patch needs a zero bound AND a first candidate in front
        // of the best one.  Flat or static content then leaves (almost) nothing instead of everything.
        const bool exact = (ub_key >> 13) == 0;
        const uint32_t ub_idx = (uint32_t)ub_key & 0x1FFFu, first_idx = (uint32_t)((q * 4 * R) * NC + prow * R);
        if constexpr (TWO) {
            const uint32_t ub256 = (uint32_t)(ub_key >> 13) << 8;          // 256 ssd < 2^32
#pragma unroll
            for (int k = 0; k < R; ++k)
                if (patch_lb[k] != 0xFFFFFFFFu && __umul24(patch_lb[k], patch_lb[k]) <= ub256 &&      // L1 <= 65280: the square fits 32 bits
                    (!exact || (patch_lb[k] == 0 && first_idx + (uint32_t)(4 * k * NC) < ub_idx))) {
                    const uint32_t slot = atomicAdd(count, 1u);
                    work[slot] = ((uint32_t)wave << 28) | ((uint32_t)lane << 22) | ((uint32_t)k << 19);
                }
        } else {
#pragma unroll
            for (int k = 0; k < R; ++k)
                if (patch_lb[k] <= ub25 && (!exact || (patch_lb[k] == 0 && first_idx + (uint32_t)(4 * k * NC) < ub_idx))) {
                    const uint32_t slot = atomicAdd(count, 1u);
                    work[slot] = ((uint32_t)wave << 28) | ((uint32_t)lane << 22) | ((uint32_t)k << 19) | (patch_lb[k] >> 6);   // NB <= 16 waves
                }
        }
        (void)ub25;
    }
    __syncthreads();
    if (d.redo_list && (int)*count > d.redo_threshold) {     // workgroup-uniform: hostile tile, brute force is cheaper
        if (tid == 0) push_redo(d, tile_id, (int)(blockIdx.x & 7));
        return true;
    }
    // ---- D2 (TWO): level 2 -- the squared bound of the listed patches, one lane per patch; survivors go to a second list at the far
    // end of `work` (the lists cannot meet while the first holds at most half of the tile's patches; a longer one is scored as it is)
    const uint32_t* list = work;
    int n = (int)*count;
    if constexpr (TWO) {
        if (2 * n <= d.nb * 64 * R) {
            for (int base = 0; base < n; base += T) {
                const int e = base + tid;
                if (e < n) {
                    const uint32_t ent = work[e];
                    const int w2 = ent >> 28, l2 = (ent >> 22) & 63, k2 = (ent >> 19) & 7;
                    const int wr2 = div_small(w2, d.magic_tc), wc2 = w2 - wr2 * d.tc;
                    const uint64_t* sp = s8 + (16 * wr2 + (l2 >> 2) * R) * XQ + wc2 * 4 + (l2 & 3) * R;
                    const uint32_t lb25 = bounds2_patch<R>(sp, XQ, k2, lds[L.own + 2 * w2], lds[L.own + 2 * w2 + 1]);
                    const unsigned long long ub2 = best[w2];
                    const uint32_t fidx = (uint32_t)(((l2 & 3) * 4 * R + 4 * k2) * NC + (l2 >> 2) * R);
                    if (lb25 <= (uint32_t)(ub2 >> 12) && ((ub2 >> 13) != 0 || (lb25 == 0 && fidx < ((uint32_t)ub2 & 0x1FFFu))))
                        work[d.nb * 64 * R - 1 - (int)atomicAdd(count + 7, 1u)] = (ent & 0xFFF80000u) | (lb25 >> 6);
                }
            }
            __syncthreads();
            n = (int)count[7];
            list = work + d.nb * 64 * R - n;                       // entries were stored downwards: any order will do
            if (tid == 0) count[0] = (uint32_t)n;                  // statistics: patches scored (count[7] is cleared with the other counters at the next tile's start)
        }
    }

    // ---- E: LPP lanes per listed patch; lane `sub` takes anchor rows AR*sub .. AR*sub+AR-1 (R+AR-1 window
    // rows), the partial dot products are added inside the quad with DPP moves.  Four lanes per patch
    // repeat some v_alignbyte work but put four times as many waves on the (long) evaluation.
    constexpr int LPP = LPPT, AR = 16 / LPP;
    const SqTable tab = sq_table(d.sqbox + (long long)pair * d.sqbox_stride, d.H, d.pitch);
    for (int base = 0; base < n; base += T / LPP) {
        const int e = base + tid / LPP;
        const int sub = lane & (LPP - 1);
        bool active = e < n;
        uint32_t ent = 0;
        if (active) {
            ent = list[e];
            active = (ent & 0x7FFFFu) <= (uint32_t)(best[ent >> 28] >> 18);          // floor(LBx / 2^11) vs ssd >> 5
        }
        const int w2 = ent >> 28, l2 = (ent >> 22) & 63, k2 = (ent >> 19) & 7;
        const int wr2 = div_small(w2, d.magic_tc), wc2 = w2 - wr2 * d.tc;         // the patch's block inside the tile
        const int prow2 = l2 >> 2, q2 = l2 & 3;
        uint32_t acc[R][4];
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) acc[i][e4] = 0;
        if (active) {
            const uint32_t* lrow = win + (16 * wr2 + prow2 * R + AR * sub) * d.pitch_dw + wc2 * 4 + q2 * R + k2;
            const uint32_t* an = anchor + w2 * ANCHOR_STRIDE + AR * 4 * sub;
            // anchor rows t - i meet window row t: each row is read ONCE, when it enters (i = 0), and stays in `keep` for the
            // R - 1 window rows that follow (round 4: the per-(t, i) reads were not merged by the compiler -- 48 b128 LDS
            // reads per patch, now 16)
            u32x4 keep[R];
#pragma unroll
            for (int t = 0; t < R + AR - 1; ++t) {
                uint32_t w[5];
#pragma unroll
                for (int s = 0; s < 5; ++s) w[s] = lrow[t * d.pitch_dw + s];
                if (t <= AR - 1) keep[t % R] = *(const u32x4*)(an + t * 4);
                u32x4 ar[R];                               // anchor rows t - i that meet this window row
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int a = t - i;
                    if (a >= 0 && a <= AR - 1) ar[i] = keep[a % R];
                }
                // one window dword at a time: its four byte alignments (bytes 4j+e4 .. 4j+e4+3) live in four
                // registers and feed the R x 4 accumulators, then the next dword reuses them
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t sh[4];
                    sh[0] = w[j];
#pragma unroll
                    for (int e4 = 1; e4 < 4; ++e4) sh[e4] = __builtin_amdgcn_alignbyte(w[j + 1], w[j], (uint32_t)e4);
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        const int a = t - i;
                        if (a < 0 || a > AR - 1) continue;
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) acc[i][e4] = __builtin_amdgcn_udot4(sh[e4], ar[i][j], acc[i][e4], false);
                    }
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) asm volatile("" : "+v"(acc[i][e4]));
                }
                // keep the rows' dot products in program order (see k_exh_dot16)
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) asm volatile("" : "+v"(acc[i][e4]));
            }
        }
        if (LPP > 1) {                                     // lane groups are uniform in `active` (one entry per group)
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    acc[i][e4] += SEA_DPP(acc[i][e4], 0xB1);                     // quad_perm [1,0,3,2]
                    if (LPP > 2) acc[i][e4] += SEA_DPP(acc[i][e4], 0x4E);        // quad_perm [2,3,0,1]
                }
        }
        if (active && sub == 0) {
            const int c02 = (bcol0 + wc2) * 16, r02 = (trow * d.tr + wr2) * 16;
            const int lo_c = max(0, d.sw - c02), hi_c = min(NC - 1, d.W - 16 - c02 + d.sw);
            const int lo_r2 = max(0, d.sw - r02), hi_r2 = min(NC - 1, d.H - 16 - r02 + d.sw);
            const bool rows_inside2 = NC == 16 * R && lo_r2 == 0 && hi_r2 == NC - 1;
            const uint32_t a2 = a2s[w2];
            unsigned long long key = ~0ull;
            const int ci0 = q2 * 4 * R + 4 * k2, ri0 = prow2 * R;
            const long long tabrow = (long long)(r02 - d.sw + ri0) * d.pitch + (c02 - d.sw + ci0);     // % 4 == 0
            if (rows_inside2 && lo_c == 0 && hi_c == NC - 1) {
                uint32_t b2[R][4];                             // all table reads in flight before the first use
#pragma unroll
                for (int i = 0; i < R; ++i) sq4(tab, tabrow + (long long)i * d.pitch, b2[i]);
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4)
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        const uint32_t cost = a2 + b2[i][e4] - 2u * acc[i][e4];
                        key = u64min_(key, ((unsigned long long)cost << 13) | (unsigned)((ci0 + e4) * NC + ri0 + i));
                    }
            } else {
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const int ci = ci0 + e4;
                    if (ci < lo_c || ci > hi_c) continue;
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        const int ri = ri0 + i;
                        if (ri < lo_r2 || ri > hi_r2) continue;
                        const uint32_t cost = a2 + sq1(tab, tabrow + (long long)i * d.pitch + e4) - 2u * acc[i][e4];
                        key = u64min_(key, ((unsigned long long)cost << 13) | (unsigned)(ci * NC + ri));
                    }
                }
            }
            if (key != ~0ull) atomicMin(&best[w2], key);
        }
        __syncthreads();
    }

    // ---- F
    if (wave_ok && lane == 0) {
        const int idx = (int)(best[wave] & 0x1FFF);
        const int ci = idx / NC, ri = idx - ci * NC;
        int32_t* o = d.mf + (((long long)pair * d.nbr + brow) * d.nbc + bcol) * 2;
        o[0] = ci - d.sw;
        o[1] = ri - d.sw;
        lds[L.prev + wave] = (uint32_t)idx;                // next tile's third probe
    }
    return false;
}

template <int R, int LPPT>
struct MseTile {
    struct Pre { uint32_t a01, a23, mine2; };
    static __device__ __forceinline__ Pre prep(const SeaDev&, uint32_t* lds, const Layout& L, int wave, int lane, bool wave_ok, uint32_t mine)
    {
        Pre p = { 0, 0, 0 };
        if (wave_ok) {
            lds[L.anchor + wave * ANCHOR_STRIDE + lane] = mine;
            anchor_quadrants(mine, &p.a01, &p.a23);
            // minus twice the sums, per half (<= 16320 each, so doubling does not carry): the addend of lower_bounds_mse's v_pk_mad_i16
            p.a01 = __builtin_bit_cast(uint32_t, -(__builtin_bit_cast(s16x2, p.a01 << 1)));
            p.a23 = __builtin_bit_cast(uint32_t, -(__builtin_bit_cast(s16x2, p.a23 << 1)));
            p.mine2 = __builtin_amdgcn_udot4(mine, mine, 0u, false);
            const uint32_t a2 = wave_sum_u32(p.mine2);
            if (lane == 0) lds[L.a2 + wave] = a2;
        }
        return p;
    }
    static __device__ __forceinline__ bool phases(const SeaDev& d, uint32_t* lds, const Layout& L, int pair, int trow, int bcol0,
                                                  uint32_t mine, const Pre& p, int tid, int tile_id)
    {
        return tile_phases_mse<R, LPPT>(d, lds, L, pair, trow, bcol0, mine, p.a01, p.a23, p.mine2, tid, tile_id);
    }
};

template <int R, int LPPT>
__global__ void __launch_bounds__(1024) k_exh_sea16_mse(SeaDev d)
{
    extern __shared__ uint32_t lds[];
    const Layout L = layout_of(d, R);
    int pair, trow, bcol0;
    if (!locate(d, &pair, &trow, &bcol0)) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // ---- A
    stage_window(d, lds + L.win, d.cur + (long long)pair * d.plane_stride, bcol0, trow * d.tr * 16);
    uint32_t mine = 0;
    const WaveBlock wb = wave_block(d, trow, bcol0, wave);
    if (wb.ok) {
        const uint8_t* aptr = d.prev + (long long)pair * d.plane_stride + (long long)(wb.brow * 16) * d.pitch + wb.bcol * 16;
        mine = *(const uint32_t*)(aptr + (long long)(lane >> 2) * d.pitch + (lane & 3) * 4);
    }
    const typename MseTile<R, LPPT>::Pre pre = MseTile<R, LPPT>::prep(d, lds, L, wave, lane, wb.ok, mine);
    if (lane == 0) lds[L.prev + wave] = (uint32_t)(d.sw * (2 * d.sw + 16) + d.sw);       // no previous tile: the zero vector
    if (threadIdx.x == 0) { lds[L.count] = 0; lds[L.count + 4] = 0; lds[L.count + 5] = 0; lds[L.count + 7] = 0; }
    __syncthreads();
    MseTile<R, LPPT>::phases(d, lds, L, pair, trow, bcol0, mine, pre, (int)threadIdx.x, tile_number(d, pair, trow, bcol0));
    if (threadIdx.x == 0) {                                // list length is final behind phase D
        atomicAdd(d.status + GME_STATUS_STATS + 16 * (blockIdx.x & 7), lds[L.count]);
        atomicAdd(d.status + GME_STATUS_STATS + 16 * (blockIdx.x & 7) + 2, lds[L.count]);
    }
}

#ifndef SEA_MSE_LPP
#define SEA_MSE_LPP 1
#endif
template <int R, int NV, int GEO = 0>
__global__ void __launch_bounds__(1024, (R <= 3 ? 8 : 6)) k_exh_sea16p_mse(SeaDev d)
{
    extern __shared__ uint32_t lds[];
    fix_geometry<R, GEO>(d);
    const Layout L = layout_of(d, R);
    if ((threadIdx.x & 63) == 0) lds[L.prev + (threadIdx.x >> 6)] = (uint32_t)(d.sw * (2 * d.sw + 16) + d.sw);    // third probe of the first tile: the zero vector
    persistent_tiles<NV, MseTile<R, SEA_MSE_LPP>>(d, lds, L);
}

}  // namespace

int launch_bbme_sea_mse(gme_ctx* ctx, const BbmeJob& job, bool* handled)
{
    *handled = false;
    if (job.procedure != GME_SEARCH_EXHAUSTIVE || job.bs != 16 || job.pnorm != GME_NORM_MSE) return GME_OK;
    if (job.sqbox_cur == nullptr || job.sw < 0 || job.sw % 4 != 0) return GME_OK;
    if (getenv("GME_FORCE_GENERIC") || getenv("GME_EXH_BRUTE")) return GME_OK;
    const int NC = 2 * job.sw + 16, R = (NC + 15) / 16;
    if (R < 1 || R > 5 || NC * NC > 8192) return GME_OK;
    const int nbr = job.H / 16, nbc = job.W / 16;
    if (nbr == 0 || nbc == 0) return GME_OK;
    SeaDev d;
    d.status = (uint32_t*)ctx->status; d.dynamic = 0;
    d.prev = job.prev; d.cur = job.cur; d.plane_stride = job.plane_stride;
    d.pairs = job.pairs; d.H = job.H; d.W = job.W; d.pitch = job.pitch; d.sw = job.sw;
    d.nbr = nbr; d.nbc = nbc; d.mf = job.mf;
    d.sqbox = job.sqbox_cur; d.sqbox_stride = job.sqbox_stride;
#ifdef GME_SEA_STAMPS
    d.stamps = nullptr;
#endif
    size_t lds = 0;
    if (!plan(R, nbr, nbc, job.sw, &d, &lds)) return GME_OK;        // does not fit: the dot4 kernel takes it
    const dim3 block(64 * d.nb);
    // hostile tiles (bound prunes little) -> brute-force redo kernel behind this one; GME_SEA_REDO=0 switches it off,
    // GME_SEA_REDO_FRAC sets the share of a tile's patches from which phase E costs more than evaluating everything
    d.redo_list = nullptr; d.redo_threshold = 0x7FFFFFFF;
    // No ordered evaluation (phase C2 of bbme_sea.hip) here: measured -6 % on every content (round 4): with one
    // lane per patch a 720x480 tile's list is a single pass either way, and under MSE the quadrant bound is too weak for the
    // tightened upper bound to prune much (pan240 x2: 30 -> 25 % of the patches).
    d.quota = 0; d.bisect = 0; d.engage = 0;
    const bool redo = !(getenv("GME_SEA_REDO") && atoi(getenv("GME_SEA_REDO")) == 0);
    if (redo) {
        int rc = ctx_redo_list(ctx, (size_t)job.pairs * d.wg_per_pair, &d.redo_list);
        if (rc) return rc;
        const double frac = getenv("GME_SEA_REDO_FRAC") ? atof(getenv("GME_SEA_REDO_FRAC")) : REDO_DEFAULT_FRAC;
        d.redo_threshold = (int)(frac * d.nb * 64 * R);
        if (!job.status_fresh) GME_HIP_TRY(hipMemsetAsync(d.status + GME_STATUS_REDO, 0, 2 * sizeof(uint32_t), ctx->stream));
    }
    const PersistPlan pp = plan_persistent(d, lds, job.pairs, ctx->prop.multiProcessorCount);
    const int nv = pp.nv;
    // The persistent form must hold the prefetched tile in registers next to phase E's 4R accumulators
    // inside the 64 VGPRs of 8 waves/SIMD: only R = 3 with 16 staging rows per thread does not fit
    // (the compiler would spill the prefetch itself) and keeps the one-tile kernel.
    const bool fits = R != 3 || nv <= 12;
    if (pp.use && fits) {
        const dim3 grid((unsigned)(8 * pp.g));
        if (pp.dynamic) {
            d.dynamic = 1;
            if (!job.status_fresh) GME_HIP_TRY(hipMemsetAsync(d.status + GME_STATUS_TILECTR, 0, 8 * 16 * sizeof(uint32_t), ctx->stream));
        }
        // the two BASELINE shapes (720x480 sw 16: 2x4 tiles; 1080p sw 32: 2x6 tiles) have instances with the tile
        // geometry folded in at compile time; GME_SEA_GENERIC=1 keeps the run-time form (A/B, tests)
        const bool fixed_ok = !getenv("GME_SEA_GENERIC");
        const bool fix3 = fixed_ok && R == 3 && nv <= 5 && geometry_matches(d, 3, 2 * 16 + 4);
        const bool fix5 = fixed_ok && R == 5 && nv <= 7 && geometry_matches(d, 5, 2 * 16 + 6);
        plan_note(ctx, (long long)job.pairs * nbr * nbc * 64 * R, "k_exh_sea16p_mse<%d,%d> tiles %dx%d persistent-%s%s grid %u lds %zu",
                  R, fix3 ? 5 : fix5 ? 7 : nv <= 6 ? 6 : nv <= 8 ? 8 : nv <= 12 ? 12 : 16, d.tr, d.tc, pp.dynamic ? "dynamic" : "static", (fix3 || fix5) ? " geometry-fixed" : "", grid.x, lds);
        if (fix3) {
            hipLaunchKernelGGL((k_exh_sea16p_mse<3, 5, 2 * 16 + 4>), grid, block, lds, ctx->stream, d);
        } else if (fix5) {
            hipLaunchKernelGGL((k_exh_sea16p_mse<5, 7, 2 * 16 + 6>), grid, block, lds, ctx->stream, d);
        } else
#define SEA_LAUNCH_P(RR, NVV) hipLaunchKernelGGL((k_exh_sea16p_mse<RR, NVV>), grid, block, lds, ctx->stream, d)
#define SEA_LAUNCH_PN(RR) do { if (nv <= 6) SEA_LAUNCH_P(RR, 6); else if (nv <= 8) SEA_LAUNCH_P(RR, 8); \
                               else if (nv <= 12) SEA_LAUNCH_P(RR, 12); else SEA_LAUNCH_P(RR, 16); } while (0)
        switch (R) {
        case 1: SEA_LAUNCH_PN(1); break;
        case 2: SEA_LAUNCH_PN(2); break;
        case 3: SEA_LAUNCH_PN(3); break;
        case 4: SEA_LAUNCH_PN(4); break;
        default: SEA_LAUNCH_PN(5); break;
        }
#undef SEA_LAUNCH_PN
#undef SEA_LAUNCH_P
    } else {
        dim3 grid;
        GME_REQUIRE(grid_for(d, &grid), GME_ERR_ARG, "too many workgroups in one launch");
    plan_note(ctx, (long long)job.pairs * nbr * nbc * 64 * R, "k_exh_sea16_mse<%d> tiles %dx%d one-tile grid %ux%ux%u lds %zu", R, d.tr, d.tc,
              grid.x, grid.y, grid.z, lds);
        const bool e4 = getenv("GME_SEA_E4") ? atoi(getenv("GME_SEA_E4")) != 0 : false;   // measured: four lanes per patch lose 9 % here (repeated v_alignbyte work)
#define SEA_LAUNCH(RR) do { if (e4) hipLaunchKernelGGL((k_exh_sea16_mse<RR, 4>), grid, block, lds, ctx->stream, d); \
                            else hipLaunchKernelGGL((k_exh_sea16_mse<RR, 1>), grid, block, lds, ctx->stream, d); } while (0)
        switch (R) {
        case 1: SEA_LAUNCH(1); break;
        case 2: SEA_LAUNCH(2); break;
        case 3: SEA_LAUNCH(3); break;
        case 4: SEA_LAUNCH(4); break;
        default: SEA_LAUNCH(5); break;
        }
#undef SEA_LAUNCH
    }
    GME_HIP_TRY(hipGetLastError());
    *handled = true;
    if (redo) return launch_exh_redo(ctx, job, R, d.tr, d.tc, d.wg_per_row, d.wg_per_pair, d.redo_list, d.status + GME_STATUS_REDO, d.status + GME_STATUS_REDO + 1);
    return GME_OK;
}
